"""softbody-webgpu_amd: MI355X-native drop-in for the physics step of spsquared/softbody-webgpu.

The directory name carries a hyphen (it is the reference's name), so load it with
`__graft_entry__.load_package()` or importlib; inside, modules use relative imports.

  layout.py   host-visible byte layouts (mirror of src/engineMapping.ts)
  scenes.py   synthetic scene generators (generalised main.ts:addRectangle)
  engine.py   ctypes binding of the C ABI in include/softbody.h (libsoftbody_hip.so)
  csrc/       HIP kernels + the C ABI + the N-API addon
  host/       JavaScript/TypeScript host mirror of engine.ts / engineMapping.ts / engineWorker.ts
"""
from . import layout, scenes  # noqa: F401
from .layout import LAYOUT_V1, LAYOUT_V2, Buffers  # noqa: F401


def __getattr__(name):
    if name in ("engine", "Engine", "EngineError", "halo"):
        import importlib
        mod = importlib.import_module(".engine" if name != "halo" else ".halo", __name__)
        return mod if name in ("engine", "halo") else getattr(mod, name)
    raise AttributeError(name)
