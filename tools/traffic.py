"""HBM bytes per frame from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc.sh.
Usage: python tools/traffic.py PMC_DIR WORKLOAD FRAMES > one.json; python tools/traffic.py merge one.json ... > profiles/traffic.json
FETCH_SIZE and WRITE_SIZE are in KiB... rocprofv3 reports them in kilobytes (1024 B); on gfx950 FETCH_SIZE
tallies 128-byte requests at 64 bytes, so it is doubled (MI355X_MICROARCH.md, section HBM)."""
import csv, glob, json, sys, collections
if sys.argv[1] == "merge":   # python tools/traffic.py merge a.json b.json ... > profiles/traffic.json (keyed by workload)
    print(json.dumps({d["workload"]: d for d in (json.load(open(f)) for f in sys.argv[2:])}, indent=1))
    sys.exit(0)
root, workload, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
tot = collections.defaultdict(float)
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not ("rt::wf_" in k or "render_kernel" in k or "path_kernel" in k):
            continue
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            per_kernel[k][r["Counter_Name"]] += float(r["Counter_Value"])
fetch = tot["FETCH_SIZE"] * 1024.0 * 2.0 / frames
write = tot["WRITE_SIZE"] * 1024.0 / frames
import ctypes, os
try:
    _L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "librt_amd.so"))
    _L.rt_version.restype = ctypes.c_char_p
    lib = _L.rt_version().decode()
except OSError:
    lib = None
out = {"workload": workload, "lib": lib, "frames": frames, "hbm_bytes_per_launch": fetch + write,
       "fetch_bytes_per_launch_corrected_x2": fetch, "write_bytes_per_launch": write,
       "launch": "one rt_render call = every kernel of one frame",
       "per_kernel_bytes_per_frame": {k: {"fetch_x2": v["FETCH_SIZE"] * 2048.0 / frames, "write": v["WRITE_SIZE"] * 1024.0 / frames}
                                      for k, v in per_kernel.items()},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py --steps 2 --warmup 1 (tools/pmc.sh); "
                 "kilobyte units, FETCH_SIZE doubled per MI355X_MICROARCH.md"}
print(json.dumps(out, indent=1))
