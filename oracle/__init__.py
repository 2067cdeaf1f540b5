"""CPU oracle for the SemStereo cost-volume + 3-D aggregation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``semstereo_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU port.

Every function is this repo's own restatement (PyTorch CPU fp32 ops, plus a
plain-C twin in ``oracle_ops.c`` for the volume/regression ops) of the
algorithm of the reference function named in its docstring
(``/root/reference/<file>:<line>``).  The reference has no tests or golden
vectors of its own, so the oracle is pinned by the
fixtures under ``tests/golden/``, which were generated in the build container
by importing the reference itself (``tests/golden/make_golden.py``).
"""
from . import detdata, ops, stack, hot_segment  # noqa: F401
