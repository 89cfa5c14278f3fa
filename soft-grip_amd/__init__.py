"""softgrip-mi355x: batched soft-gripper simulator (MI355X-native hot path of mbed92/soft-grip)."""
import importlib

from .mjcf import Model, compile_mjcf, load_model  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not need torch or the HIP library
    if name in ("ManEnv", "Env", "SimulationError"):
        return getattr(importlib.import_module(__name__ + ".manenv"), name)
    if name in ("native", "manenv", "create_dataset", "build_native"):
        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
