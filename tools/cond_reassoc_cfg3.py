"""dev (GPU): the re-association probes of tests/test_gpu_headline_parity.py::test_conditioned_sequence_survives_reassociation at the
BENCHMARKED shape -- all 50 frames of cfg3_50c under CODD_OPT_GN_Q4 = 256 and under the heuristic launch configurations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_headline_parity as T

for tag, kw in (("gn_q4_256", dict(options=dict(gn_q4=256))), ("heuristic_launch_configurations", dict(shipped_tuning=False))):
    rows = T.run_conditioned("cfg3_50c", **kw)
    T.assert_rule1(f"cfg3_50c[{tag}]", rows)
    print(f"cfg3_50c[{tag}]: worst frame {max(r['mean'] for r in rows):.2e} px, worst flipped fraction {max(r['flipped'] for r in rows):.1e}")
