"""Builders for the BASELINE.json configurations (synthetic where the reference checkout lacks
the asset).  Pure host code (numpy)."""
import numpy as np

from .scene import (Camera, Cube, Material, Mesh, MeshObject, RenderSettings, Scene, Sphere, Texture,
                    default_material)


def config1(width=256, height=256, spp=1, depth=4):
    """C1: default Cube (identity) + Sphere at (1.5,0,0), both `Default` material,
    object order [cube, sphere]; camera (0,0,10), identity rotation, FOV 45, exposure 1; sky
    defaults; AA on; frameCount 0."""
    scene = Scene()
    mat = default_material()
    cube = Cube()
    cube.setMaterial(mat)
    sphere = Sphere(position=(1.5, 0.0, 0.0))
    sphere.setMaterial(mat)
    scene.addToScene("cube", cube)
    scene.addToScene("sphere", sphere)
    cam = Camera((0.0, 0.0, 10.0), (0.0, 0.0, 0.0, 0.0), 45.0, 1.0)
    cam.setWidth(width)
    cam.setHeight(height)
    cam.calcRays()
    st = RenderSettings(bounceCount=depth, sampleCount=spp)
    return scene, cam, st
