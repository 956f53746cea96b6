"""raytracer_amd -- MI355X-native render path behind the API of rmehri01/raytracer.

The package is a thin Python face over two in-tree native libraries (see _native.py).
Rendering happens only in the HIP kernels of csrc/; importing fails if they are not built.
"""
from ._native import RenderError, RtCameraDesc, RtSceneDesc, RtStats  # noqa: F401
from .api import (Camera, Canvas, Compound, Conic, DeviceScene, FlatScene, Material, Matrix, Operation,  # noqa: F401
                  Pattern, PointLight, Primitive, Shape, World, EXERCISES, exercise, exercise_shading, obj, synthetic)
