"""Guard on the built code objects of libggnn_amd.so (no GPU needed: the kernel descriptors' metadata
is read with llvm-objdump / llvm-readelf from the ROCm image).

Why: the traversal kernels live on memory-latency hiding.  Any scratch (private segment) traffic in
their pop loop costs a `s_waitcnt vmcnt(0)` per reload -- i.e. a wait for the rows just requested --
and its stores are real HBM writes.  Round 5 lost 17 % of the headline kernel to an array of three
candidate keys that the optimiser had moved to scratch WITHOUT reporting a spilled register (DESIGN.md
section 4, "Waits, read in the ISA" (iv)): the default kernels of the four BASELINE shapes must keep a
private segment of zero bytes and their register budget (72 VGPRs = 7 waves per SIMD for the float
kernels, 64 = 8 waves for uint8 rows)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ggnn_amd", "csrc", "libggnn_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"

pytestmark = pytest.mark.skipif(
    not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))),
    reason="needs the built library and the ROCm llvm tools")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    """{demangled kernel name: {metadata key: int}} over every gfx950 code object in the library."""
    d = tmp_path_factory.mktemp("co")
    lib = shutil.copy(LIB, d)   # llvm-objdump writes the extracted bundles next to its input
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], cwd=d, check=True,
                   capture_output=True)
    out, names = {}, []
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)],
                               capture_output=True, text=True, check=True).stdout
        cur = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.(\w+):\s+(\S+)\s*$", line)
            if not m:
                continue
            key, val = m.groups()
            if key == "name" and val.startswith("_Z"):
                cur = out.setdefault(val, {})
                names.append(val)
            elif cur is not None and key in ("private_segment_fixed_size", "vgpr_count",
                                             "vgpr_spill_count", "sgpr_spill_count"):
                cur[key] = int(val)
    assert names, "no kernels found in the library's code objects"
    demangled = subprocess.run(["c++filt"], input="\n".join(out), capture_output=True, text=True,
                               check=True).stdout.splitlines()
    return {re.sub(r"\(.*", "", n).replace("ggnn_amd::", "").replace("void ", ""): v
            for n, v in zip(demangled, out.values())}


# (kernel, VGPR budget): the variants bench.py's four shapes and the default build launch
# (profiles/r05*_kernel_stats.csv name them)
DEFAULT_KERNELS = [
    # 1M x 128 f32 (headline), hashed set with one / two bucket registers, ring-less, early rows
    ("query_kernel<float, 16, 2, 1, 0, Prescreen<8, 1, 0>, 1, true, true>", 72),
    ("query_kernel<float, 16, 2, 1, 0, Prescreen<8, 1, 0>, 2, true, true>", 72),
    # 12.5M x 96 f32 shard
    ("query_kernel<float, 8, 3, 1, 0, Prescreen<8, 1, 0>, 1, true, true>", 72),
    ("query_kernel<float, 8, 3, 1, 0, Prescreen<8, 1, 0>, 2, true, true>", 72),
    # long searches on the ring-less tag set
    ("query_kernel<float, 16, 2, 1, 0, Prescreen<8, 1, 0>, -8, true, true>", 72),
    ("query_kernel<float, 16, 2, 1, 0, Prescreen<8, 1, 0>, -9, true, true>", 72),
    # uint8 rows read directly
    ("query_kernel<unsigned char, 8, 1, 1, 0, NoPrescreen, 1, true, true>", 64),
    ("query_kernel<unsigned char, 8, 1, 1, 0, NoPrescreen, 2, true, true>", 64),
    # construction: the counting merge kernels (tests, the build roofline) ...
    ("merge_kernel<float, 16, 2, 1, 0, Prescreen<8, 1, 0>, 1, true, true>", 80),
    ("merge_kernel<float, 8, 3, 1, 0, Prescreen<8, 1, 0>, 1, true, true>", 80),
    ("merge_kernel<unsigned char, 8, 1, 1, 0, NoPrescreen, 1, true, true>", 72),
    ("sym_kernel<float, 16, 2, 1, 0, NoPrescreen>", 80),
    ("sym_kernel<float, 8, 3, 1, 0, NoPrescreen>", 80),
    ("sym_kernel<unsigned char, 8, 1, 1, 0, NoPrescreen>", 72),
]


@pytest.mark.parametrize("name,vgprs", DEFAULT_KERNELS)
def test_default_kernels_have_no_scratch(kernels, name, vgprs):
    assert name in kernels, f"{name} is not in the library (renamed template parameters?)"
    k = kernels[name]
    assert k["private_segment_fixed_size"] == 0, k
    assert k["vgpr_spill_count"] == 0, k
    assert k["vgpr_count"] <= vgprs, k


# ... and the ones a production build launches (COUNT = false, round 6).  The float variants keep
# three registers in scratch ACROSS THE LAYER LOOP (stored in front of it, reloaded in
# SortedList::transform, i.e. once per layer of a point -- read in the ISA: merge.hip:116-117),
# nothing inside the pop loop; the bound below is that state, so that any growth shows up here.
@pytest.mark.parametrize("name,vgprs,scratch", [
    ("merge_kernel<float, 16, 2, 1, 0, Prescreen<8, 1, 0>, 1, true, false>", 80, 16),
    ("merge_kernel<float, 8, 3, 1, 0, Prescreen<8, 1, 0>, 1, true, false>", 80, 16),
    ("merge_kernel<unsigned char, 8, 1, 1, 0, NoPrescreen, 1, true, false>", 72, 0)])
def test_production_merge_kernels_keep_their_scratch_bound(kernels, name, vgprs, scratch):
    assert name in kernels, f"{name} is not in the library (renamed template parameters?)"
    k = kernels[name]
    assert k["private_segment_fixed_size"] <= scratch, k
    assert k["vgpr_count"] <= vgprs, k


def test_every_early_rows_query_kernel_of_an_l2_base_is_scratch_free(kernels):
    """All `EARLY = true` squared-L2 query variants (whatever ring home / bucket count the launcher
    picks): no private segment.  (Cosine variants carry a second accumulator and may spill a few
    registers: not a BASELINE shape, listed by scripts/kernel_resources.py.)"""
    seen = 0
    for name, k in kernels.items():
        m = re.match(r"query_kernel<(float|unsigned char), \d+, \d+, 1, 0, .*, true, (true|false)>$",
                     name)
        if not m or not name.endswith("true, true>"):
            continue
        seen += 1
        assert k["private_segment_fixed_size"] == 0, (name, k)
    assert seen >= 8
