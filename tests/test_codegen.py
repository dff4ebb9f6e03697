"""The generated blocks of the segment trainer's step (qampy_amd/csrc/train_seg_blocks*.inc) are what scripts/gen_seg_blocks.py writes today -
the generator also checks every statement against the wait states gfx950 does not interlock (a packed result is not read by the next
instruction, a VALU result not through DPP by the next two) - and its checker does reject a schedule that breaks them."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "scripts", "gen_seg_blocks.py")


def test_generated_blocks_are_in_sync_with_the_generator(tmp_path):
    subprocess.check_call([sys.executable, GEN, str(tmp_path)])
    for name in ("train_seg_blocks.inc", "train_seg_blocks_ba.inc"):
        with open(os.path.join(ROOT, "qampy_amd", "csrc", name)) as a, open(tmp_path / name) as b:
            assert a.read() == b.read(), "%s is stale: run scripts/gen_seg_blocks.py" % name


def test_wait_state_checker_rejects_hazards():
    spec = importlib.util.spec_from_file_location("gen_seg_blocks", GEN)
    g = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, [GEN]
    try:
        spec.loader.exec_module(g)
    finally:
        sys.argv = argv
    dpp = "v_add_f32_dpp v254, v254, v254 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
    with pytest.raises(AssertionError):
        g.check(["v_pk_mul_f32 v[248:249], %[x0], %[w0]", "v_pk_add_f32 v[250:251], v[248:249], v[252:253]"], "packed result read by the next instruction")
    with pytest.raises(AssertionError):
        g.check(["v_pk_add_f32 v[254:255], v[248:249], v[250:251]", "ds_read_b128 %[d0], %[la] offset:%[o0]", dpp], "one wait state before a DPP read")
    g.check(["v_pk_add_f32 v[254:255], v[248:249], v[250:251]", "ds_read_b128 %[d0], %[la] offset:%[o0]", "s_nop 0", dpp], "two wait states")
    for tpl in (4, 6):
        for nl in (1, 2):
            for nr in range(4):
                for keep in (0, 1):
                    g.check(g.block(tpl, keep, nl, nr), "block A")
                for d1 in (0, 1):
                    g.check(g.block_b(tpl, d1) + g.block(tpl, 1, nl, nr), "blocks B + A")
