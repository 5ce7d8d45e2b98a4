"""Where the RCCL stand-in of tests/ddp/fake_rccl.cpp lives (built on demand when hipcc is here)."""
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("_build_fake_rccl", os.path.join(_HERE, "ddp", "build_fake.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
FAKE_RCCL = _mod.build()
