"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/confignet_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "confignet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from confignet_amd import _lib
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(_lib.lib, n), "libconfignet_hip.so does not export %s" % n
    assert _lib.lib.cn_version() >= 1
    # every declared int-returning function has a ctypes signature (binding == header)
    assert set(_lib.SIGNATURES) | {"cn_last_error_string"} == set(names)


def test_bad_arguments_return_error_codes_not_crashes():
    from confignet_amd import _lib
    g = _lib.CnConvGeom()
    rc = _lib.lib.cn_conv_fwd(ctypes.byref(g), None, None, None, None, 0, 0.0, None)
    assert rc == -1
    assert b"nd must be" in _lib.lib.cn_last_error_string()
    assert _lib.lib.cn_gemm(0, 0, 0, 1, 1, None, 1, None, 1, None, 1, None, 0, 0.0, None) == -1


def test_geometry_same_padding_rules():
    from confignet_amd.ops import ConvSpec
    g = ConvSpec((4, 4), up=1).geom((2, 16, 16, 64), 32)
    assert (g.out_h, g.out_w, g.p_h, g.p_w, g.up) == (32, 32, 1, 1, 1)
    g = ConvSpec((3, 3), stride=2).geom((2, 256, 256, 3), 48)
    assert (g.out_h, g.p_h, g.s_h) == (128, 0, 2)
    g = ConvSpec((3, 3, 3), up=1).geom((2, 4, 4, 4, 512), 256)
    assert (g.out_d, g.out_h, g.out_w, g.p_d) == (8, 8, 8, 1)
    g = ConvSpec((7, 7), stride=2, explicit_pad=3).geom((1, 256, 256, 3), 64)
    assert (g.out_h, g.p_h) == (128, 3)


def test_no_register_of_an_in_flight_lds_read_is_touched_before_its_wait(tmp_path):
    """The LDS-DMA kernels (wgrad2, fwd2, Winograd F(2x2) / F(4x4)) read their MFMA operands with ds_read instructions inside inline asm
    and wait for them with hand-counted s_waitcnt lgkmcnt(n): the compiler does not know the asm's result arrives later, so any
    copy it places between the asm and the wait reads the register too early (round 4: tied "+v" operands of the wait became
    v_mov copies in wgrad2_kernel<128x128> -- tile-shaped errors and NaN under LDS contention, never in isolation).
    scripts/isa_lds_hazard.py replays the compiled ISA (hipcc cross-compiles here) and must find no such access."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import isa_lds_hazard
    csrc = os.path.join(ROOT, "confignet_amd", "csrc")
    for name in ("wgrad2", "winograd", "winograd4", "fwd2"):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-save-temps", "-c",
                        os.path.join(csrc, name + ".hip"), "-I" + csrc, "-I" + os.path.join(ROOT, "include"), "-o", name + ".o"],
                       cwd=tmp_path, check=True, capture_output=True)
        isa = os.path.join(tmp_path, name + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        assert os.path.exists(isa)
        text = open(isa).read()
        assert "ds_read" in text and "lds" in text                       # (the check below is not vacuous)
        sys.argv = ["isa_lds_hazard.py", isa]
        assert isa_lds_hazard.main() == 0, "a ds_read destination is read before its s_waitcnt in " + name


def test_the_image_kernels_issue_all_their_staging_loads_before_the_first_wait(tmp_path):
    """c3_fwd_kernel / c7s2_fwd_kernel (3x3 and 7x7 convolutions of the 3-channel image) stage an input patch in LDS.  Written as
    `patch[i] = in_image ? X[...] : 0` in a loop, hipcc gave every guarded load its own exec-mask region and an `s_waitcnt vmcnt(0)`
    in front of the LDS store: thirteen dependent memory round trips per workgroup, 27 us of lifetime around 1.5 us of MFMA work
    (round 5: 213 -> 118 us once the loads were batched; one instance still kept seven of the waits until the loads became
    branch-free).  The compiled ISA of every instance must show NO wait for vmcnt(0) that is followed by another global load
    before the workgroup's first barrier."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "confignet_amd", "csrc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-save-temps", "-c",
                    os.path.join(csrc, "igemm_conv.hip"), "-I" + csrc, "-I" + os.path.join(ROOT, "include"), "-o", "igemm_conv.o"],
                   cwd=tmp_path, check=True, capture_output=True)
    lines = open(os.path.join(tmp_path, "igemm_conv-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*(c3_fwd_kernel|c7s2_fwd_kernel)\S*:", l)]
    assert len(starts) >= 12                                        # 4 + 4 instances of the 3x3 kernel, 2 + 2 of the 7x7 one
    for i in starts:
        j = i
        while "s_endpgm" not in lines[j]:
            j += 1
        body = lines[i:j]
        k = next(n for n, l in enumerate(body) if "s_barrier" in l)
        pre = body[:k]
        loads = [n for n, l in enumerate(pre) if "global_load_dword" in l]
        assert len(loads) >= 17, lines[i]                           # filter slice + one patch element per row
        early = [n for n, l in enumerate(pre) if re.search(r"s_waitcnt.*vmcnt\(0\)", l) and n < loads[-1]]
        assert not early, "%s: %d waits for vmcnt(0) with staging loads still to be issued" % (lines[i].split(":")[0], len(early))
