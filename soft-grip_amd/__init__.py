"""softgrip-mi355x: batched soft-gripper simulator (MI355X-native hot path of mbed92/soft-grip)."""
from .mjcf import Model, compile_mjcf, load_model  # noqa: F401
