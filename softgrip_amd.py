"""Importable alias for the ``soft-grip_amd`` package directory (a hyphen is not a valid identifier)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("soft-grip_amd")
