"""no GPU needed: the host-side integer contracts of libconp_hip.so against the CPU oracle --
k-vector tables (bit-exact, incl. ug), the GPU plan's (planar, kz, sign) decomposition, electrode index bookkeeping over a
re-neighbour with re-ordered atoms, and the electrode-row regrouping of LAMMPS half lists (newton on/off, etypes or generic)."""
import numpy as np
import pytest

import oracle_py
from conp_amd import capi, neighbor, systems

CASES = {
    "dilute_ffield": lambda: systems.deck("dilute", "ffield"),
    "dilute_slab": lambda: systems.deck("dilute", "slab"),
    "il_onelayer_ffield": lambda: systems.deck("il_onelayer", "ffield"),
    "il_onelayer_slab": lambda: systems.deck("il_onelayer", "slab"),
    "il_twolayer_noslab_zneutr": lambda: systems.deck("il_twolayer", "noslab_zneutr"),
    "headline": lambda: systems.synthetic_fast(),
}


@pytest.mark.parametrize("case", list(CASES))
def test_ktables_bit_exact_vs_oracle(oracle, case):
    s = CASES[case]()
    kt = capi.host_ktables(s)
    ks = oracle_py.KSpace.from_system(oracle, s)
    for name in ("kcount", "kcount_flat", "kcount_expand", "kxmax", "kymax", "kzmax", "kmax", "kmax3d"):
        assert kt[name] == getattr(ks, name), name
    assert list(kt["kcount_dims"]) == list(ks.kcount_dims)
    for name in ("kxvecs", "kyvecs", "kzvecs", "kxy_list", "kz_list", "ug"):
        assert np.array_equal(kt[name], getattr(ks, name)), name
    # SURVEY section 8 sizes
    if case == "headline":
        assert (kt["kcount"], kt["kcount_flat"]) == (99773, 645)
    if case == "il_onelayer_ffield":
        assert (kt["kcount"], kt["kcount_flat"]) == (4668, 134)
    # the plan's decomposition reproduces every k vector: k = (planar p, sign * m)
    kz = kt["plan_sign"] * kt["plan_m"]
    assert np.array_equal(kz, ks.kzvecs)
    p = kt["plan_p"]
    assert p.min() == 0 and p.max() == kt["n_planar"] - 1
    for pid in np.unique(p)[:50]:
        sel = p == pid
        assert len(set(zip(ks.kxvecs[sel], ks.kyvecs[sel]))) == 1      # one (kx, ky) per planar index
    ks.close()


def test_index_bookkeeping_bit_exact_vs_oracle(oracle):
    s = systems.deck("il_onelayer", "ffield", shuffle_seed=3)
    at, alist, blist = neighbor.build_lists(s)
    fo = oracle_py.Fix(oracle, s)
    fo.set_atoms(at); fo.set_lists(alist, blist); fo.post_neighbor()
    m0 = fo.maps()
    h0 = capi.host_index(at.tag[:at.nlocal], at.echeck[:at.nlocal])
    for k in ("ele2tag", "ele2eleall", "eleall2tag", "eleall2ele", "elebuf2eleall", "tag2eleall"):
        assert np.array_equal(h0[k], m0[k]), k
    assert list(h0["sizes"][:3]) == [832, 832, s.natoms - 832]
    # re-neighbour with re-ordered atoms: permanent numbering must survive, volatile maps follow the new order
    perm = np.random.default_rng(1).permutation(s.natoms)
    s2 = s.copy(); s2.x, s2.q, s2.type, s2.tag, s2.echeck = s.x[perm], s.q[perm], s.type[perm], s.tag[perm], s.echeck[perm]
    at2, alist2, blist2 = neighbor.build_lists(s2)
    fo.set_atoms(at2); fo.set_lists(alist2, blist2); fo.post_neighbor()
    m1 = fo.maps()
    h1 = capi.host_index(at.tag[:at.nlocal], at.echeck[:at.nlocal], at2.tag[:at2.nlocal], at2.echeck[:at2.nlocal])
    for k in ("ele2tag", "ele2eleall", "eleall2tag", "eleall2ele", "elebuf2eleall", "tag2eleall"):
        assert np.array_equal(h1[k], m1[k]), k
    assert np.array_equal(h1["eleall2tag"], h0["eleall2tag"]) and not np.array_equal(h1["ele2eleall"], h0["ele2eleall"])
    fo.close()


def _reference_b_pairs(at, lst, newton, tag2eleall):
    """the membership rule of blist_coul_cal (fix_conp.cpp:1326-1350), pair by pair, in plain Python"""
    rows = []
    for i in lst.ilist:
        eci = at.echeck[i] != 0
        for j in lst.neigh[lst.first[i]: lst.first[i] + lst.numneigh[i]] & neighbor.NEIGHMASK:
            ecj = at.echeck[j] != 0
            if (eci ^ ecj) and (newton or eci or j < at.nlocal):
                if eci:
                    rows.append((tag2eleall[at.tag[i]], i, j))
                elif j < at.nlocal or newton:
                    rows.append((tag2eleall[at.tag[j]], j, i))
    return rows


@pytest.mark.parametrize("newton", [False, True])
@pytest.mark.parametrize("etypes", [True, False])
def test_pair_rows_match_the_reference_membership_rule(newton, etypes):
    s = systems.small_random(ne_side=4, n_elyte=96, lz=60.0, mode="slab")
    s.newton = newton
    if not etypes:
        s.eletypes = None
    at, alist, blist = neighbor.build_lists(s, special_frac=0.1)
    h = capi.host_index(at.tag[:at.nlocal], at.echeck[:at.nlocal])
    got = capi.host_pair_rows(1, blist, at, newton)
    ref = _reference_b_pairs(at, blist, newton, h["tag2eleall"])
    assert got["npairs"] == len(ref)
    ne = int(h["sizes"][1])
    for row in range(ne):                              # same pairs per row, list order kept inside the row
        mine = list(zip(got["ele_atom"][got["row_ptr"][row]: got["row_ptr"][row + 1]],
                        got["oth_atom"][got["row_ptr"][row]: got["row_ptr"][row + 1]]))
        theirs = [(e, o) for r, e, o in ref if r == row]
        assert mine == theirs, row
    # every physical electrode-electrolyte pair within the neighbour cutoff appears exactly once (newton on or off)
    phys = {(int(at.tag[e]), int(at.tag[o]), tuple(np.round(at.x[e] - at.x[o], 6))) for e, o in zip(got["ele_atom"], got["oth_atom"])}
    assert len(phys) == got["npairs"]
    # a-list: ele-ele pairs, single-count rule of fix_conp.cpp:1268
    ga = capi.host_pair_rows(0, alist, at, newton)
    assert ga["npairs"] > 0 and np.all(at.echeck[ga["ele_atom"]] != 0) and np.all(at.echeck[ga["oth_atom"]] != 0)
    assert np.array_equal(ga["col"], h["tag2eleall"][at.tag[ga["oth_atom"]]])
    # post-force pair set: no newton / ghost filter (fix_conp.cpp:1411)
    gp = capi.host_pair_rows(2, blist, at, newton)
    assert gp["npairs"] >= got["npairs"]


def test_sk_gemm_chunk_loop_does_not_spill():
    """the dominant kernel sits at the 256-register budget of two waves per SIMD; a change that tips the NFW = 5 bodies over it
    makes the allocator spill inside the chunk loop (round 3: -15 % at the 16384 / 262144 size, invisible in the parity tests).
    Cross-compile the kernels for gfx950 (no GPU needed), read the compiler's resource remarks and the assembly: no scratch
    access and no SGPR spill traffic (v_readlane / v_writelane) in any loop that multiplies (a basic block of a loop whose blocks
    hold MFMA instructions).  The epilogue of a segment (sk_project_out, sk_sum_pieces) may park a few of the NEXT segment's
    constants: once per segment, outside the chunk loop -- bounded here."""
    import os
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "lammps-user-conp2_amd", "csrc", "conp_kernels.hip")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        p = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", asm,
                            "-I" + os.path.join(root, "include"), "-Rpass-analysis=kernel-resource-usage"],
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        text = open(asm).read()
    blocks = re.split(r"remark: Function Name: ", p.stderr)
    # (the instantiation every size but the small decks runs: sk_gemm_kernel<false>; <true> adds the fused phase prologue)
    sk = [b for b in blocks if b.startswith("_ZN4conp14sk_gemm_kernelILb0E")]
    assert len(sk) == 1
    scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", sk[0]).group(1))
    sspill = int(re.search(r"SGPRs Spill: (\d+)", sk[0]).group(1))
    vgprs = int(re.search(r" VGPRs: (\d+)", sk[0]).group(1))
    assert sspill <= 32 and scratch <= 128, (scratch, sspill)
    assert vgprs <= 256
    # the kernel's body, cut into basic blocks with the loop each belongs to (the compiler's own annotations)
    m = re.search(r"^_ZN4conp14sk_gemm_kernelILb0E\w*:[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
    assert m
    loop_of, body_of, cur = {}, {}, None
    lines = m.group(1).split("\n")
    for k, ln in enumerate(lines):
        lab = re.match(r"^(\.LBB\d+_\d+):(.*)$", ln)
        if lab:
            cur = lab.group(1)
            body_of[cur] = []
            note = lab.group(2) + " " + (lines[k + 1] if k + 1 < len(lines) else "")
            hdr = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", note)
            if hdr:
                loop_of[cur] = (hdr.group(1), int(hdr.group(2)))
            elif "This Inner Loop Header" in note or "This Loop Header" in note:
                d = re.search(r"Header: Depth=(\d+)", note)
                loop_of[cur] = (cur[2:], int(d.group(1)) if d else 1)
        elif cur:
            body_of[cur].append(ln)
    mfma_loops = {loop_of[b][0] for b, body in body_of.items()
                  if b in loop_of and loop_of[b][1] >= 2 and any("v_mfma_f64" in x for x in body)}
    assert mfma_loops, "no chunk loop found"
    spill_ops = ("scratch_", "v_readlane_b32", "v_writelane_b32")      # VGPR spills; SGPR spills go through VGPR lanes
    bad = [b for b, body in body_of.items()
           if b in loop_of and loop_of[b][0] in mfma_loops and any(op in x for x in body for op in spill_ops)]
    assert not bad, bad

