"""CPU: the C-ABI library builds/loads and exports every symbol include/moyolo.h declares."""
import ctypes
import os
import re

import pytest

from mo_yolo_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "moyolo.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(moy_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in moyolo.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes SIGNATURES drifted from the header"


def _header_fields(struct):
    txt = open(os.path.join(ROOT, "include", "moyolo.h")).read()
    body = txt[txt.index(f"typedef struct {struct} {{"):txt.index(f"}} {struct};")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        if "," in decl:
            head, *rest = decl.split(",")
            names.append(head.replace("*", " ").split()[-1])
            names += [r.strip().lstrip("*").strip() for r in rest]
        else:
            names.append(decl.replace("*", " ").split()[-1])
    return names


@pytest.mark.parametrize("struct,mirror", [("moy_gemm_args", "GemmArgs"), ("moy_c2f_args", "C2fArgs"), ("moy_decoder_tail_args", "DecoderTailArgs"),
                                           ("moy_decoder_mid_args", "DecoderMidArgs"), ("moy_msda_raw_args", "MsdaRawArgs")])
def test_args_layout_matches_header(struct, mirror):
    """Field order of every ctypes mirror == field order of its struct in include/moyolo.h (round 5: w_packed / wc_packed appended)."""
    # arrays: `int32_t run_tok0[4]` in the header <-> ("run_tok0", c_int32 * 4) in the mirror
    want = []
    for f in getattr(_lib, mirror)._fields_:
        n_el = getattr(f[1], "_length_", None)
        want.append(f"{f[0]}[{n_el}]" if n_el else f[0])
    assert _header_fields(struct) == want


def test_pack_mfma_a_is_the_layout_the_header_states():
    """ops.pack_mfma_a against the definition in include/moyolo.h (above moy_decoder_tail_args): the 16-byte piece of lane (r, q), half j,
    32-row group g, 32-column panel pn sits at byte (((g * (K/32) + pn) * 2 + j) * 64 + lane) * 16 and holds W[32 g + 16 j + r][32 pn + 8 q .. + 7]."""
    import torch
    from mo_yolo_amd.ops import pack_mfma_a
    N, K = 96, 128
    W = torch.arange(N * K, dtype=torch.int32).view(N, K).to(torch.int16)        # distinct 16-bit values
    P = pack_mfma_a(W).reshape(-1)
    for g, pn, j, lane in [(0, 0, 0, 0), (2, 3, 1, 63), (1, 2, 0, 17), (2, 0, 1, 48), (0, 3, 1, 5)]:
        r, q = lane & 15, lane >> 4
        off = (((g * (K // 32) + pn) * 2 + j) * 64 + lane) * 8                      # in 16-bit elements
        assert torch.equal(P[off:off + 8], W[32 * g + 16 * j + r, 32 * pn + 8 * q:32 * pn + 8 * q + 8])
    assert sorted(P.tolist()) == sorted(W.reshape(-1).tolist())                   # a permutation


def test_version_and_errors_no_gpu_needed():
    lib = _lib.lib()
    assert lib.moy_version() == 100
    assert b"invalid" in lib.moy_strerror(-22)
    # argument validation happens on the host before any launch: NULL args -> EINVAL
    assert lib.moy_gemm(None, None) == -22
    assert lib.moy_sigmoid_f32(None, 0, None, None) == -22


def test_host_side_contracts_of_the_round4_entries_no_gpu_needed():
    """moy_set_cu_limit is host state (per thread, returns the previous budget); a folded 1x1 consumer (`post_*`) on a launch that
    has no fused form is refused with MOY_ENOSYS on the host -- never computed without the consumer."""
    import ctypes as C
    lib = _lib.lib()
    assert lib.moy_set_cu_limit(128) == 0 and lib.moy_set_cu_limit(0) == 128 and lib.moy_set_cu_limit(-5) == 0 and lib.moy_set_cu_limit(0) == 0
    a = _lib.GemmArgs()
    fake = 0x10000                       # aligned, never dereferenced: the refusal comes before any launch
    a.A, a.lda, a.W, a.C, a.ldc = fake, 128, fake, fake, 128
    a.M, a.N, a.K, a.ksize, a.stride, a.dtype = 1 << 17, 128, 128, 1, 1, _lib.BF16
    a.post_W, a.post_n, a.post_act = fake, 128, 1
    assert lib.moy_gemm(C.byref(a), None) == _lib.ENOSYS


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MoyoloError):
        _lib.lib()


def test_cpu_tensors_are_rejected_like_the_reference_op():
    import torch
    from mo_yolo_amd import ops
    v = torch.zeros(1, 4, 1, 4)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.ms_deform_attn_forward(v, torch.tensor([[2, 2]]), torch.tensor([0]), torch.zeros(1, 1, 1, 1, 1, 2),
                                   torch.zeros(1, 1, 1, 1, 1))


def test_gemm_query_answers_the_dispatch_on_the_host():
    """moy_gemm_query (round 5): validation + kernel eligibility of a real moy_gemm call, no launch, no device access -- so it runs
    here without a GPU.  The engine's planner asks it instead of trial-launching on uninitialised buffers."""
    lib = _lib.lib()
    buf = (ctypes.c_char * 4096)()
    al = (ctypes.addressof(buf) + 63) // 64 * 64                    # a 64-byte aligned address: the query dereferences nothing

    def q(**kw):
        a = _lib.GemmArgs()
        a.A, a.W, a.C, a.lda, a.ldc, a.ksize, a.stride, a.dtype = al, al, al, kw.get("K", 256), kw.get("N", 256), 1, 1, _lib.BF16
        for k, v in kw.items():
            setattr(a, k, v)
        k = ctypes.c_int(-7)
        rc = lib.moy_gemm_query(ctypes.byref(a), ctypes.byref(k))
        return rc, k.value

    assert q(M=300, N=256, K=256) == (0, _lib.KERNEL_TILED)                        # the decoder's query-sized linears
    assert q(M=70000, N=256, K=256, c_rows_per_batch=1000, c_batch_stride=1300) == (0, _lib.KERNEL_WREG)   # input_proj with its token scatter
    assert q(M=70000, N=1536, K=128, ldc=32, plane_cols=32, plane_stride=70000 * 32) == (0, _lib.KERNEL_WREG)   # the P3 value form
    assert q(M=70000, N=256, K=256, dtype=_lib.F32) == (0, _lib.KERNEL_TILED)        # fp32: always the exact tiled kernel
    assert q(M=100000, N=512, K=1024) == (0, _lib.KERNEL_DMA)                       # deep K: the 256 x 256 LDS-DMA tiles
    # a plane of more than 1 GiB is outside the 32-bit lane offsets of the value form: the tiled kernel takes the launch instead
    assert q(M=70000, N=1536, K=128, ldc=32, plane_cols=32, plane_stride=1 << 30) == (0, _lib.KERNEL_TILED)
    # a form the arguments REQUIRE (row runs: score pass over the valid rectangle) below the persistent kernel's launch size: refused
    runs = dict(run_levels=1, run_period=1000, run_tok0=(ctypes.c_int32 * 4)(0), run_pitch=(ctypes.c_int32 * 4)(10), run_len=(ctypes.c_int32 * 4)(10),
                run_rows=(ctypes.c_int32 * 4)(10))
    rc, k = q(M=5000, N=256, K=256, C=None, ln_g=al, ln_b=al, dot_w=al, dot_b=al, dot_out=al, dot_n=1, **runs)
    assert (rc, k) == (_lib.ENOSYS, 0)
    assert q(M=300, N=250, K=256)[0] == _lib.EINVAL                                  # N % 4
    geom = dict(ksize=3, stride=2, B=64, Hin=152, Win=272, Hout=76, Wout=136, Cin=64, M=64 * 76 * 136, N=128, K=576, lda=64, ldc=128, act=_lib.ACT_SILU,
                scale=al, shift=al)
    assert q(**geom) == (0, _lib.KERNEL_CONV_S2)
    assert q(**dict(geom, post_W=al, post_scale=al, post_shift=al, post_n=128, post_act=_lib.ACT_SILU)) == (0, _lib.KERNEL_CONV_S2)
    small = dict(geom, B=1, M=76 * 136)
    assert q(**small)[0] == 0 and q(**small)[1] != _lib.KERNEL_CONV_S2
    assert q(**dict(small, post_W=al, post_scale=al, post_shift=al, post_n=128, post_act=_lib.ACT_SILU))[0] == _lib.ENOSYS   # never computed without its consumer


def test_product_library_reads_no_environment_and_the_lab_lives_in_its_own_build():
    """Round 6 (SURVEY section 8(b) 'no global state'): the kernels' sources call getenv nowhere (`knob()` of common.hpp is a
    compile-time constant in the product build and an environment read only under -DMOY_DIAG=1); the shipped libmoyolo.so holds none of
    the `MOY_*` switch names and none of the timing-only / stamped template instances; the lab build (libmoyolo_diag.so) has both and the
    same C ABI; the host code reads no `MOY_*` variable outside `PlanOptions.from_env` (lab) and the bench's own launcher settings."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, "mo_yolo_amd", "csrc")
    for f in glob.glob(os.path.join(csrc, "*.hip")):
        assert "getenv" not in open(f).read(), f
    hpp = open(os.path.join(csrc, "common.hpp")).read()
    assert hpp.count("getenv") == 1 and hpp.index("#if MOY_DIAG") < hpp.index("getenv") < hpp.index("#else")
    prod, lab = build.build(verbose=False), build.build(verbose=False, diag=True)
    bp, bl = open(prod, "rb").read(), open(lab, "rb").read()
    for name in (b"MOY_GEMM_WREG", b"MOY_WREG_ABL", b"MOY_CWS_VARIANT", b"MOY_TILE", b"MOY_CU_LIMIT", b"MOY_SL1_DIAG", b"MOY_TAIL_ABL", b"MOY_MR_MFMA"):
        assert name not in bp, name
        assert name in bl, name
    assert len(bp) < 0.75 * len(bl), (len(bp), len(bl))            # the lab's extra instances are most of a third of it
    nm = lambda p: {ln.split()[-1] for ln in subprocess.run(["nm", "-D", "--defined-only", p], capture_output=True, text=True, check=True).stdout.splitlines()
                    if " T " in ln and ln.split()[-1].startswith("moy_")}
    assert nm(prod) == nm(lab) == set(header_symbols())
    # timing-only / stamped kernel instances: only the lab has them (device code is embedded in the .so: search the mangled names)
    for frag in (b"decoder_tail_kernelINS_6bf16_tELi2E", b"stem_l1_kernelINS_6bf16_tELi2E", b"conv_ws_pp_kernel"):
        assert frag not in bp and frag in bl, frag
    # host side: MOY_* variables are read by PlanOptions.from_env (lab) and by nothing else in the package
    pkg = os.path.join(ROOT, "mo_yolo_amd")
    for f in glob.glob(os.path.join(pkg, "*.py")):
        txt = open(f).read()
        if os.path.basename(f) == "engine.py":
            body = txt[:txt.index("    def from_env(")] + txt[txt.index("class View"):]
            assert "os.environ" not in body, "engine.py reads the environment outside PlanOptions.from_env"
        elif os.path.basename(f) not in ("_lib.py", "build.py"):
            assert "os.environ" not in txt, f


def test_kernel_resource_check_fails_a_ring_kernel_with_scratch():
    """build.py parses hipcc's kernel-resource-usage remarks; scratch in an LDS-DMA ring kernel (hand-counted vmcnt waits) is a
    build error, scratch elsewhere is listed."""
    rem = """x.hip:1:1: remark: Function Name: _ZN3moy16gemm_wreg_kernelINS_6bf16_tELi32EEEvNS_10WregParamsE [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     ScratchSize [bytes/lane]: 24 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark: Function Name: _ZN3moy19decoder_tail_kernelINS_6bf16_tELi0EEEv21moy_decoder_tail_args [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     ScratchSize [bytes/lane]: 88 [-Rpass-analysis=kernel-resource-usage]
"""
    table, errors = build.check_resources("x.hip", rem)
    assert len(table) == 2 and len(errors) == 1 and "gemm_wreg_kernel" in errors[0] and "24 bytes/lane" in errors[0]
    res = os.path.join(ROOT, "mo_yolo_amd", "csrc", "obj", "resources.txt")
    build.build(verbose=False)
    if os.path.exists(res):       # (absent on a box that received the prebuilt library without the object directory)
        rows = open(res).read().splitlines()
        assert len(rows) > 200
        assert not [r for r in rows if any(k in r for k in build.RING_KERNELS) and " scratch    0 " not in r]
