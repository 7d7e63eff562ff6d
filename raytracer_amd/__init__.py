"""raytracer_amd -- MI355X-native per-pixel render path behind the reference's Scene / Camera /
DisplayObject / Material API.  Host mirror: scene.py / hostmath.py; C-ABI binding: capi.py;
HIP kernels + C-ABI implementation: csrc/."""
