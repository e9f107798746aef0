"""Static check of the compiled gfx950 code (no GPU): no 16-byte store may have its data registers overwritten by the VALU instruction(s)
right behind it.

Round 4 root-caused the "wrong y beside a second process" failures of conv_chain_kernel<true, .> to exactly that sequence:
`buffer_store_dwordx4 v[24:27], v160, s[48:51], s14 offen` / `v_mov_b32 v24, v16`.  The store reads its data over several cycles after it
issues; hipcc's hazard recogniser pads the global / flat / soffset-0 forms with two wait states but exempts MUBUF stores with an SGPR soffset,
and on MI355X that exemption does not hold (scripts/vmem_store_war_probe.hip, profiles/r04_store_war_probe.txt: lanes 12-15 of every row
pick up the new contents; one wait state is enough for that form, two for the soffset-0 form).  csrc/conv_chain.hip now pins wait states
behind its stores (store16); this test keeps any kernel of the library from growing the exposure back.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stmask_amd", "csrc")


def _listings():
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8", "isa"])
    d = os.path.join(CSRC, "isa")
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".s"))


def test_no_wide_store_has_its_data_overwritten_within_two_wait_states():
    files = _listings()
    assert len(files) >= 12
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "lint_store_war.py"), *files, "--window", "3", "--min-wait", "2"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-4000:]


def test_the_lint_sees_the_round3_sequence(tmp_path):
    """The checker must flag the sequence that shipped in round 3 (and accept the padded one), or the test above proves nothing."""
    bad = tmp_path / "bad.s"
    bad.write_text("k:\n\tbuffer_store_dwordx4 v[24:27], v160, s[48:51], s14 offen\n\tv_mov_b32_e32 v24, v16\n\ts_endpgm\n")
    good = tmp_path / "good.s"
    good.write_text("k:\n\tbuffer_store_dwordx4 v[24:27], v160, s[48:51], s14 offen\n\ts_nop 3\n\tv_mov_b32_e32 v24, v16\n"
                    "\tglobal_store_dwordx4 v[0:1], v[4:7], off\n\tv_add_u32_e32 v0, 1, v0\n\ts_endpgm\n")
    lint = os.path.join(ROOT, "scripts", "lint_store_war.py")
    assert subprocess.run([sys.executable, lint, str(bad)], capture_output=True).returncode == 1
    assert subprocess.run([sys.executable, lint, str(good)], capture_output=True).returncode == 0


# Kernels whose ring protocols count vector-memory operations by hand (`s_waitcnt vmcnt(N)` with N derived from the DMAs / gathers the source issues):
# a compiler-inserted scratch spill or reload is a vector-memory operation nobody counted, and the protocol would then certify a slab that has
# not landed -- silently.  Their code objects must therefore use no private segment at all (and the other hot kernels none either: a scratch
# reload is a vmcnt(0)-class wait behind every store in flight, csrc/conv_chain.hip's history).
COUNTED_WAIT_KERNELS = {"dcn_fused.s": ("dcn_fused_kernel",), "conv_chain.s": ("conv_chain_kernel",), "conv_kxr.s": ("conv_kxr_kernel",),
                        "conv_bf16x.s": ("conv_planar_kernel", "conv_planar_kx3_kernel"), "stem_fused.s": ("stem_fused_kernel",)}


def _kernel_metadata(path):
    """(.name, {field: int}) for every kernel of the listing's amdhsa.kernels metadata."""
    out, cur = [], None
    for line in open(path):
        t = line.strip()
        if t.startswith(".name:") and "_Z" in t:
            cur = (t.split(":", 1)[1].strip(), {})
            out.append(cur)
        elif cur is not None and ":" in t and t.split(":", 1)[0] in (".private_segment_fixed_size", ".sgpr_spill_count", ".vgpr_spill_count"):
            cur[1][t.split(":", 1)[0]] = int(t.split(":", 1)[1])
    return out


def test_counted_wait_kernels_use_no_scratch():
    files = {os.path.basename(f): f for f in _listings()}
    seen = 0
    for fname, kernels in COUNTED_WAIT_KERNELS.items():
        for name, meta in _kernel_metadata(files[fname]):
            if not any(k in name for k in kernels):
                continue
            seen += 1
            # (SGPR spills go to VGPR lanes -- v_writelane / v_readlane, no memory operation -- and are not counted by vmcnt)
            assert meta.get(".private_segment_fixed_size") == 0 and meta.get(".vgpr_spill_count") == 0, (name, meta)
    assert seen >= 20        # (8 dcn_fused + 3 conv_chain + the conv_kxr / conv_planar instantiations)
