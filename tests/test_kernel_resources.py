"""CPU: what the compiler made of the hot kernels, read from the gfx950 code object inside libsmr_hip.so (no GPU needed).

A kernel that silently starts to use scratch memory is several times slower on the MI355X and nothing else tells: during round 2 a
by-reference lambda in k_seed_pg's string loop put the search's ranges into 152 bytes of private memory per lane and tripled the kernel's
time; results, tests and the emulator were all unaffected.  The limits below are the register / scratch budgets DESIGN.md argues with."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from sortmerna_amd import build

LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_metadata():
    bundler, readelf = os.path.join(LLVM, "clang-offload-bundler"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(bundler) and os.path.exists(readelf) and shutil.which("objcopy")):
        pytest.skip("ROCm LLVM tools not installed")
    lib = build.build_library()
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fatbin"), os.path.join(d, "gfx950.o")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([bundler, "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
        notes = subprocess.check_output([readelf, "--notes", co]).decode()
    out = {}
    for blk in re.split(r"\n  - \.agpr_count", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        f = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))   # noqa: E731
        out[name] = dict(vgpr=f("vgpr_count"), spill=f("vgpr_spill_count"), scratch=f("private_segment_fixed_size"), lds=f("group_segment_fixed_size"))
    return out


def _kernel_isa(*parts):
    """disassembly lines of the kernel whose mangled name contains every part"""
    bundler, objdump = os.path.join(LLVM, "clang-offload-bundler"), os.path.join(LLVM, "llvm-objdump")
    if not (os.path.exists(bundler) and os.path.exists(objdump) and shutil.which("objcopy")):
        pytest.skip("ROCm LLVM tools not installed")
    lib = build.build_library()
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fatbin"), os.path.join(d, "gfx950.o")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([bundler, "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
        text = subprocess.check_output([objdump, "-d", co]).decode()
    out, on = [], False
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            on = all(p in m.group(1) for p in parts)
            continue
        if on and line.strip():
            out.append(line.split("//")[0].strip())
    assert out, "kernel %s not in the code object" % (parts,)
    return out


def _find(md, *parts):
    hits = [k for k in md if all(p in k for p in parts)]
    assert hits, "kernel %s not in the code object" % (parts,)
    return [md[k] for k in hits]


def test_hot_kernels_use_no_scratch_and_stay_within_their_register_budget():
    md = _kernel_metadata()
    # seed stage: everything in registers; the search at (almost) full occupancy
    for parts, max_vgpr in [(("k_seed_keys",), 64), (("k_seed_cscan",), 64), (("k_seed_split",), 128),      # (round 6: pieces of 16 384 tuples = 140 KB of LDS = one 1024-thread block per CU = 4 waves per SIMD: 128 registers are its budget)
                             (("k_seed_bins",), 128),       # (round 6: pieces of 16 384 tuples = 128 KB of LDS = one block per CU like the first pass, sixteen loads per thread in flight: 124 registers of its 128)
                            (("k_seed_pgILi0",), 72), (("k_seed_pgILi1",), 72), (("k_seed_finish",), 72), (("k_seed_searchILi",), 64),
                            (("k_candE",), 64), (("k_trace_bandILi8",), 128), (("k_trace_bandILi16",), 128), (("k_trace_wide",), 64)]:
        for k in _find(md, *parts):
            assert k["scratch"] == 0 and k["spill"] == 0, (parts, k)
            assert k["vgpr"] <= max_vgpr, (parts, k)
    # static LDS: only what the source declares.  (A per-read struct indexed with a run-time value is moved to LDS by the compiler -- 48 bytes per
    # thread, 12 KB per block of k_cand and k_seed_finish in round 2 -- and nothing but this number tells.)
    for parts, max_lds in [(("k_candE",), 1024 + 64),        # (round 6: + the list of the block's 256 reads that go on to phase 2)
                            (("k_seed_keys",), 64), (("k_seed_pgILi",), 64), (("k_seed_finish",), 3 * 8 * 256 * 4 + 64)]:
        for k in _find(md, *parts):
            assert k["lds"] <= max_lds, (parts, k)
    # k_chain is built for 3 waves per SIMD (168 VGPRs) and is allowed the spills DESIGN.md 3.2 accounts for
    for k in _find(md, "k_chainILb"):                 # <EXT, LONG>; the LONG instantiations (batches with reads of several strips) carry more state
        assert k["vgpr"] <= 168 and k["spill"] <= 300, k
    for k in _find(md, "k_chainILb0ELb0"):            # the instantiation of the short-read workloads
        assert k["spill"] <= 200, k
    # the stand-alone Smith-Waterman / begin-cell kernels: 4 waves per SIMD
    for parts in [("k_ssw_batch",), ("k_beginsILb0",)]:
        for k in _find(md, *parts):
            assert k["vgpr"] <= 128 and k["spill"] <= 2, (parts, k)


def test_the_round_kernels_of_the_candidate_walk():
    """smr_walk.hpp.  k_walk: no Smith-Waterman registers (<= 102 VGPRs = 5 waves per SIMD), no vector spills; k_sw16<R>: the rows take every
    register its occupancy allows, and what the compiler parks in scratch are loop-invariant values around the step loops -- no scratch access
    may sit between the packed maxima of a step loop (that would be a spill per DP step; it happened with R = 19 under a 128-VGPR bound)."""
    md = _kernel_metadata()
    for k in _find(md, "k_walkILb0"):
        assert k["vgpr"] <= 102 and k["spill"] <= 4 and k["scratch"] <= 64, k
    for k in _find(md, "k_wnext") + _find(md, "k_wlist"):
        assert k["vgpr"] <= 64 and k["spill"] == 0 and k["scratch"] == 0, k
    for parts, max_vgpr in [(("k_sw16ILi13",), 128), (("k_sw16ILi19",), 168), (("k_sw16ILi32",), 256)]:
        for k in _find(md, *parts):
            assert k["vgpr"] <= max_vgpr and k["spill"] <= 96, (parts, k)
    isa = _kernel_isa("k_sw16ILi19")
    # stretches of instructions without a scratch access; every packed maximum must lie in a long one (a step loop of 4 steps x 19 rows)
    stretch, in_long, total = 0, 0, 0
    runs = []
    for ins in isa + ["scratch_end"]:
        if ins.startswith("scratch_"):
            runs.append(stretch); stretch = 0
        elif ins.startswith("v_pk_max_i16"):
            stretch += 1
    total = sum(runs)
    in_long = sum(r for r in runs if r >= 4 * 19 * 3)
    assert total >= 4 * 4 * 19 * 3 and in_long == total, (total, in_long, [r for r in runs if r])


def test_lds_leaves_room_for_the_waves_the_registers_allow():
    """Round 5's lesson: `k_seed_pg` ran at 5 waves per SIMD with a 7-wave register budget because its LDS was sized for the batch's
    longest hit list.  160 KB of LDS per CU, four SIMDs; a kernel's LDS per wave must let as many waves in as its registers do."""
    md = _kernel_metadata()
    lds_cu = 160 * 1024
    src = open(os.path.join(build.CSRC, "smr_seed_pg.hpp")).read()
    assert re.search(r"#define PG_LDS_WORDS\(ccap\) \(4u \* \(ccap\) \+ 128u\)", src) and re.search(r"#define PG_CAND_CAP0 256u", src) and re.search(r"#define PG_OCC 7\b", src)
    for k in _find(md, "k_seed_pgILi"):
        per_wave = 4 * (4 * 256 + 128) + k["lds"]                        # dynamic (PG_LDS_WORDS at the initial candidate budget) + static
        assert per_wave * 7 * 4 <= lds_cu and k["vgpr"] <= 72, k          # 7 waves per SIMD: 512 / 7 = 73 registers
    for k in _find(md, "k_walkILb0"):
        assert k["lds"] <= 7680 and k["lds"] * 5 * 4 <= lds_cu and k["vgpr"] <= 96, k        # 5 waves per SIMD; 8 160 bytes measured 6 % slower than 7 424 (profiles/r5s33_*)
    for k in _find(md, "k_seed_finish"):
        assert k["lds"] * 7 <= lds_cu and k["vgpr"] <= 72, k              # 7 blocks of four waves


def test_loads_that_belong_together_are_issued_together():
    """Round 6 ("What the ISA showed"): `if (i < np) { load; atomic }` unrolled twelve times compiled to load, s_waitcnt vmcnt(0), ds_add_rtn twelve
    times over -- twelve round trips to memory one after the other per piece of the first sort pass, and nothing but the disassembly told; k_sw16 read the
    letters of its rows with 4 R conditional loads per pass.  The shape of the fixed code is pinned here."""
    def longest_run(isa, op):
        best = run = 0
        for ins in isa:
            if ins.startswith(op):
                run += 1; best = max(best, run)
            elif ins.startswith("s_waitcnt") and "vmcnt(0)" in ins:          # (a wait that drains the queue; vmcnt(n > 0) leaves the younger loads in flight)
                run = 0
        return best
    split = _kernel_isa("k_seed_split")
    assert longest_run(split, "global_load_dwordx2") >= 16, "the sixteen tuple loads of a piece no longer leave together"
    assert not any(i.startswith("global_load_ushort") for i in split), "a load of blockDim.x (and its wait) is back in k_seed_split"
    bins = _kernel_isa("k_seed_bins")
    assert longest_run(bins, "global_load_dwordx2") >= 16
    sw = _kernel_isa("k_sw16ILi19")
    n_loads = sum(1 for i in sw if i.startswith("global_load"))
    assert n_loads <= 120 and any(i.startswith("global_load_dwordx3") for i in sw), n_loads      # (320 loads when every row asked for its letter by itself)
    cand = _kernel_isa("k_candE")
    assert longest_run(cand, "global_load_dwordx2") >= 4                                        # the four hit words of a lane
