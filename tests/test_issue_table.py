"""tools/isa/issue_table.py -- the one definition of the kernels' vector-issue roof -- on the library as built:
every extractor kernel is found in the gfx950 code object, priced with the measured per-opcode costs, and the committed table
(profiles/r06_issue_table.json, what bench.py falls back to on a host without llvm-objdump) is the table of THESE sources."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "isa"))


def test_measured_costs_cover_the_common_opcodes():
    import issue_table as IT
    m = IT.measured_costs()
    assert len(m) > 50
    assert 2.0 < m["v_add_u32"] < 3.5 and 4.0 < m["v_perm_b32"] < 5.5 and m["v_fma_f64"] > 16
    assert IT.opcode_cost("v_cndmask_b32_e32", m) == (4.5, "class:other")     # the microbenchmark's VCC stall is not an issue cost
    assert IT.opcode_cost("v_max_u16_e32", m)[1] == "measured"


def test_issue_table_of_the_built_library():
    import issue_table as IT
    if not os.path.exists(os.path.join(IT.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    t = IT.issue_table()
    k = t["kernels"]
    names = list(k)
    for prefix in ("k_pyr_fused", "k_fast_cells<128, 13>", "k_octree<false, 512>", "k_orient_blur_desc<0, false, false>",
                   "k_bow_rank_fold", "k_search_bow", "k_bfknn2_frames_mfma"):
        assert any(n.startswith(prefix) for n in names), prefix
    assert not any(n.startswith("k_fast_runs") or n.startswith("k_octree<false, 1024>") or n.startswith("k_copy_out") for n in names)
    fast = k[[n for n in names if n.startswith("k_fast_cells<128, 13>")][0]]
    assert 250 < fast["static_valu_instructions"] < 400 and 2.8 < fast["cycles_per_instruction"] < 3.8
    assert fast["share_priced_by_measurement"] > 0.5
    committed = json.load(open(os.path.join(ROOT, "profiles", "r06_issue_table.json")))
    for n in ("k_pyr_fused", "k_fast_cells<128, 13>", "k_octree<false, 512>", "k_orient_blur_desc<0, false, false>"):
        assert abs(committed["kernels"][n]["cycles_per_instruction"] - k[n]["cycles_per_instruction"]) < 0.02, n
        assert committed["kernels"][n]["static_valu_instructions"] == k[n]["static_valu_instructions"], n
