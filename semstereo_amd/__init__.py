"""semstereo_amd -- MI355X-native (gfx950) cost-volume + 3-D aggregation hot path of SemStereo.

  ops       drop-in callables with the reference's names (models/submodule.py)
  modules   nn.Module twins of the 3-D stack with the reference's state_dict keys
  segment   HotSegment: features -> disparities (models/SemStereo.py:273-323)
  install   install(model_module) / accelerate(model): drop-in into the reference's own model
  dist      one-process-per-GPU batch sharding (RCCL / gloo)

The compute lives in csrc/libsemstereo_hip.so behind the C ABI of include/semstereo_hip.h.
Nothing here falls back to the CPU or to the test oracle.
"""
from . import _lib, dist, engine, modules, ops, ops_unsigned, segment, train_layers  # noqa: F401
from .install import accelerate, install, restore_forward, uninstall  # noqa: F401
from .segment import GraphedSegment, HotSegment, PairPipeline  # noqa: F401

__version__ = "0.1.0"
