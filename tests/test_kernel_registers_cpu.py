"""Register budget of the hot kernel instantiations, checked at compile time (no GPU): no scalar register spilled to a VGPR lane (DESIGN.md 4.7;
VERDICT r4 item 7), and the vector-register spills where round 5 left them.  One device-only compile of a single instantiation per case
(scripts/hot_probe.py: a few seconds each)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

CASES = [
    # instantiation, second translation unit (-DFW_PHASE_TU), most VGPR spills allowed
    ("fw_example_kernel_r<300, true, 20, true, 1, 4>", False, 1),   # config C, store policy 4 (the shipped default since round 6)
    ("fw_example_kernel_r<300, true, 20, true, 1, 3>", False, 5),   # config C, store policy 3 (round 5's; an A/B policy since round 6: 5 spilled vector registers with the asm-issued parked rows, FW_PARK_ASM)
    ("fw_example_kernel_r<300, true, 20, true, 1, 1>", False, 4),   # config C, round 4's store policy
    ("fw_example_kernel_r<300, true, 0, true, 2, 4>", False, 0),    # two-chunk rows (k = 16), updating
    ("fw_example_kernel_r<100, false, 0, false, 2, 4>", False, 0),  # ... predict-only (config E's batched head path)
    ("fw_example_kernel<4, 300, true, 0, true>", False, 0),         # config E: the generic kernel with the deep head (in-order launches; FWGPU_NN_V2=0)
    # config E's concurrent launches since round 6: the head as a phase of the two-chunk instantiation.  No vector register spilled; 33-35 scalars live in VGPR lanes
    # (LDS offsets of the head's scratch across its loops) -- a fused kernel, which has never shown round 3's two-queue fault (tests/test_gpu_overlap.py runs this shape)
    ("fw_example_kernel_r<300, true, 0, true, 2, 4, true>", False, 0, 40),
    ("fw_example_kernel<4, 100, false, 1, false>", True, 0),        # FWD phase
    ("fw_example_kernel<4, 300, true, 3, false>", True, 0),         # UPD phase
    ("fw_example_kernel<4, 300, true, 3, true>", True, 0),          # UPD phase behind the mini-batched head
]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_no_scalar_register_is_spilled_to_a_vgpr_lane(case, tmp_path):
    kernel, phase_tu, max_vgpr_spills = case[:3]
    max_sgpr_spills = case[3] if len(case) > 3 else 0
    env = dict(os.environ)
    if phase_tu:
        env["PHASE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "hot_probe.py"), kernel], capture_output=True, text=True, env=env, timeout=600).stdout
    sg = re.search(r"SGPRs Spill: (\d+)", out)
    vg = re.search(r"VGPRs Spill: (\d+)", out)
    vr = re.search(r"\bVGPRs: (\d+)", out)
    assert sg and vg and vr, out
    assert int(sg.group(1)) <= max_sgpr_spills, out
    assert int(vg.group(1)) <= max_vgpr_spills, out
    assert int(vr.group(1)) <= 128, out
