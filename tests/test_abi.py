"""C-ABI checks that need no GPU: the library builds, loads, exports every symbol that
include/yat_hip.h declares, and the ctypes binding agrees with the header's parameter lists."""
import ctypes as C
import os
import re

from yat_amd import lib as ylib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "yat_hip.h")

_CT = {
    "int": C.c_int, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "float": C.c_float, "double": C.c_double,
    "const char*": C.c_char_p,
}


def _parse_header():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|uint64_t|const char\s*\*)\s*(yat_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, params = re.sub(r"\s+", "", m.group(1)).replace("constchar*", "const char*"), m.group(2), m.group(3).strip()
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        types = []
        for p in plist:
            if "*" in p or "yat_stream_t" in p:
                types.append("ptr")
            else:
                base = p.replace("const", "").split()[0]
                types.append(base)
        decls[name] = (ret, types)
    return decls


def test_header_declares_what_we_bind():
    decls = _parse_header()
    assert set(decls) == set(ylib.SIGNATURES), set(decls) ^ set(ylib.SIGNATURES)
    for name, (ret, types) in decls.items():
        res, args = ylib.SIGNATURES[name]
        assert res is _CT[ret], name
        assert len(args) == len(types), (name, len(args), len(types))
        for i, (a, t) in enumerate(zip(args, types)):
            if t == "ptr":
                assert a in (C.c_void_p, C.c_char_p) or issubclass(a, C._Pointer), (name, i, a)
            else:
                assert a is _CT[t], (name, i, a, t)


def test_library_loads_and_exports_every_symbol(built_lib):
    lib = ylib.load()
    assert lib.yat_version() >= 1
    for name in _parse_header():
        assert hasattr(lib, name), name


def test_workspace_queries_are_pure_host_code(built_lib):
    lib = ylib.load()
    assert lib.yat_colsum_workspace_bytes(8192, 11200) > 0
    assert lib.yat_ln_bwd_workspace_bytes(8192, 2240, 1024) == 8 * 16 * 2 * 2240 * 4      # one partial row pair per 64-row workgroup
    assert lib.yat_linear_attn_workspace_bytes(8, 1024, 70) == 8 * 70 * 33 * 32 * 4 * (2 + 4)
    assert lib.yat_gradnorm_workspace_bytes(1 << 20, 7) == 7 * 4 * 4


def test_bad_arguments_are_rejected_without_a_gpu(built_lib):
    lib = ylib.load()
    # argument validation happens before any launch, so it is testable on the CPU box
    assert lib.yat_gemm_bf16(0, 0, 0, 8, 8, None, 8, None, 8, None, 8, None, None) == -1
    assert lib.yat_gemm_bf16(0, 0, 8, 8, 12, 1, 16, 1, 16, 1, 8, None, None) == -1  # K % 8 != 0
    assert lib.yat_adamw_step(7, 1, 1, 1, 1, None, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1, 1, None, 0.0, 0, None) == -1
    assert lib.yat_ln_modulate_fwd(4, 7, 4, 1e-6, 1, 1, 1, 8, 1, 1, 1, None) == -1     # D % 8 != 0


def test_no_kernel_uses_scratch(built_lib):
    """Compiler remarks collected by the build (yat_amd/build/resources.json): no kernel of the library may spill VGPRs
    or touch scratch memory -- on this path a spill in a shared epilogue costs ~10 % of the step and no error."""
    import json
    import os
    path = os.path.join(os.path.dirname(built_lib), "build", "resources.json")
    if not os.path.exists(path):           # library built before the remarks were recorded: rebuild records them
        from yat_amd.build import build
        build(force=True, verbose=False)
    with open(path) as f:
        res = json.load(f)
    assert len(res) >= 60, "resource remarks missing"
    from yat_amd.build import SOURCES
    seen = {k.split(":")[0] for k in res}
    assert seen >= set(SOURCES) - {"comm.hip", "plan.hip"}, f"no resource remarks for {set(SOURCES) - seen}"   # (no kernels in those two)
    for name, r in res.items():
        assert r.get("ScratchSize [bytes/lane]", 0) == 0, (name, r)
        assert r.get("VGPRs Spill", 0) == 0, (name, r)
    gemm = {k: v for k, v in res.items() if "gemm256_kernel" in k}
    assert len(gemm) >= 10 and all(v["VGPRs"] + v.get("AGPRs", 0) <= 256 for v in gemm.values())     # 2 waves per SIMD


def test_bad_arguments_of_the_widened_rows_are_rejected(built_lib):
    """PixArt glue and adapter entry points: validation precedes any launch (fake non-null pointers are never dereferenced)."""
    lib = ylib.load()
    assert lib.yat_patch_rearrange(1, 4, 7, 8, 2, 1, 1, 1, 2, None) == -1                       # H % p != 0
    assert lib.yat_patch_rearrange(1, 4, 8, 8, 2, 1, 1, 1, 1, None) == -1                       # in place
    assert lib.yat_add_pos_embed(16, 4, 12, 1, 1, 1, None) == -1                                # D % 8 != 0
    assert lib.yat_ddpm_add_noise(2, 64, 1, 1, None, 1, 1, None) == -1                          # null coefficient
    assert lib.yat_mse_bf16_chunk(2, 64, 32, 1, 1, 1.0, 1, None, 1, None) == -1                 # stride < used
    assert lib.yat_dropout(64, 1.0, 7, 0, 1, 1, None) == -1                                     # p must be < 1
    assert lib.yat_rank_expand(16, 60, 8, 1, 1, 1, 64, 1.0, 0, None) == -1                      # N % 8 != 0
    assert lib.yat_rank_expand(16, 64, 12, 1, 1, 1, 64, 1.0, 0, None) == -1                     # R must be 8 or 16
    assert lib.yat_lokr_rows(16, 136, 8, 0, 1, 1, 1, None) == -1                                # N > 128
    assert lib.yat_lokr_small_wgrad(16, 8, 64, 9, 1, 1, 64, 1, 64, 1.0, 0, 1, None) == -1       # r_out > R
    assert lib.yat_lokr_small_wgrad(16, 8, 64, 8, 1, 1, 32, 1, 64, 1.0, 0, 1, None) == -1       # ldx < N
    assert lib.yat_lokr_small_wgrad_workspace_bytes(32768, 8, 2240) == 32 * 8 * 18 * 128 * 4
    assert lib.yat_sdpa_fwd(1, 64, 64, 1, 136, 0.1, 1, 136, 1, 1, 136, 1, None, 1, 136, None, None) == -1   # dh > 128


def test_gemm_epilogue_is_versioned_by_size(built_lib):
    """yat_gemm_epilogue.struct_size: the library copies exactly what the caller owns.  The full layout and the first
    published one (64 bytes) are accepted; zero, a truncated, an oversized or a misaligned size is YAT_EINVAL before anything
    is read or launched -- a binding written against an older header can never make the library read past its object."""
    lib = ylib.load()
    full = C.sizeof(ylib.GemmEpilogue)
    assert lib.yat_gemm_epilogue_size() == full == 152          # round 5 appended a2 / b2 / k2 / a2_group_n to the 128-byte layout
    assert ylib.GemmEpilogue().struct_size == full

    def call(ep):
        # M = 0 is rejected AFTER the epilogue is not looked at; use valid dims + fake non-null pointers and a bad activation
        # so that a struct that passes the size check fails later with the same code but a different cause is excluded:
        return lib.yat_gemm_bf16(0, 0, 8, 8, 8, 1, 8, 1, 8, 1, 8, C.byref(ep), None)

    ok = ylib.GemmEpilogue(None, None, 7)                       # activation 7: invalid -> EINVAL from the field check
    assert call(ok) == -1
    # accepted sizes get past validation: without a GPU the call then fails in the HIP runtime (a positive hipError_t)
    import torch
    if not torch.cuda.is_available():
        assert call(ylib.GemmEpilogue()) > 0
        for older in (64, 128):                                 # the first published layout, and the one before round 5's fields
            v = ylib.GemmEpilogue()
            v.struct_size = older
            assert call(v) > 0
    for bad in (0, 8, 56, 60, full + 8, 0x7fff0000):
        ep = ylib.GemmEpilogue()
        ep.struct_size = bad
        assert call(ep) == -1, bad

    class OldEpilogue(C.Structure):                             # the round-1 INTEGRATION.md stub: no size field, 9 members
        _fields_ = [("bias", C.c_void_p), ("aux_out", C.c_void_p), ("activation", C.c_int), ("gate", C.c_void_p),
                    ("residual", C.c_void_p), ("ld_aux", C.c_int), ("ld_gate", C.c_int), ("ld_residual", C.c_int),
                    ("rows_per_batch", C.c_int)]
    for bias in (None, 0x7f0000001000):
        old = OldEpilogue(bias, None, 0, None, None, 0, 0, 0, 0)
        rc = lib.yat_gemm_bf16(0, 0, 8, 8, 8, 1, 8, 1, 8, 1, 8, C.cast(C.byref(old), C.POINTER(ylib.GemmEpilogue)), None)
        assert rc == -1


def test_comm_entry_points_without_a_communicator(built_lib):
    """The communication quartet (+ helpers) is exported and refuses work before yat_comm_init; nothing here touches a GPU
    or loads RCCL."""
    lib = ylib.load()
    assert lib.yat_comm_world() == 0 and lib.yat_comm_rank() == -1
    assert lib.yat_bucket_allreduce_async(1, 128, 0, None, None) == -2          # YAT_ENOCOMM
    assert lib.yat_comm_wait(-1, None) == -2
    assert lib.yat_comm_broadcast(1, 128, 0, None) == -2
    assert lib.yat_comm_destroy() == 0                                           # idempotent
    assert lib.yat_comm_init(0, 0, b"x" * 128) == -1 and lib.yat_comm_init(2, 2, b"x" * 128) == -1
    assert lib.yat_comm_init(0, 1, None) == -1 and lib.yat_comm_unique_id(None) == -1
    assert isinstance(lib.yat_comm_last_error(), bytes)
    # round 6: the sharded optimizer step's two collectives
    assert lib.yat_bucket_reduce_scatter_async(1, 128, 0, None, None) == -2
    assert lib.yat_comm_allgather(1, 128, None) == -2


def test_gradnorm_pieces_argument_checks(built_lib):
    """The piece form of the clip norm (round 6, include/yat_hip.h): argument errors are refused before any launch -- no GPU
    needed -- and the host-side piece table (yat_amd/optim.py norm_pieces) cuts tensors exactly at the eighths of their buckets."""
    lib = ylib.load()
    assert lib.yat_gradnorm_pieces_partial(None, 4, 1, 1, 1, None, 1, None) == -1            # no gradient buffer
    assert lib.yat_gradnorm_pieces_partial(1, 0, 1, 1, 1, None, 1, None) == -1               # no pieces
    assert lib.yat_gradnorm_pieces_partial(1, 70000, 1, 1, 1, None, 1, None) == -1           # more pieces than grid.y holds
    assert lib.yat_gradnorm_pieces_partial(1, 4, 1, 1, 0, None, 1, None) == -1               # no chunk
    assert lib.yat_gradnorm_pieces_partial(1, 4, 1, 1, 1, None, None, None) == -1            # no output
    assert lib.yat_gradnorm_pieces_finish(0, 1, 1, 1, 1.0, 1, 1, None) == -1                 # no tensors
    assert lib.yat_gradnorm_pieces_finish(3, 1, 1, 1, 1.0, None, 1, None) == -1              # no norm output
    from yat_amd.optim import NORM_CHUNK, norm_pieces
    seg = [0, 1000 * 64, 1000 * 64 + 8, 9000 * 64]                    # three tensors; the middle one is a bias of 8 elements
    buckets = [(0, 1000 * 64), (1000 * 64, 9000 * 64)]
    ps, tf, cb, mx, part = norm_pieces(seg, buckets)
    assert ps[0] == 0 and ps[-1] == seg[-1] and all(b > a for a, b in zip(ps, ps[1:]))
    assert tf == [0, 8, 9, len(ps) - 1]                                # tensor 0 = 8 pieces (it IS bucket 0), the bias 1, the rest of bucket 1
    for bi, (lo, hi) in enumerate(buckets):
        e = (hi - lo) // 8
        assert all(lo + k * e in ps for k in range(8))                 # every eighth starts a piece
    assert all(0 <= k < 8 for _, k in part)
    assert cb[-1] == sum((b - a + NORM_CHUNK - 1) // NORM_CHUNK for a, b in zip(ps, ps[1:])) and mx == max(
        (b - a + NORM_CHUNK - 1) // NORM_CHUNK for a, b in zip(ps, ps[1:]))
    # a bucket that is not 8 x 16-byte parts is not cut (and cannot be sharded: part -1)
    ps2, tf2, _, _, part2 = norm_pieces([0, 100, 200], [(0, 200)])
    assert ps2 == [0, 100, 200] and tf2 == [0, 1, 2] and part2 == [(0, -1), (0, -1)]


def test_integration_stub_matches_the_header():
    """INTEGRATION.md's Level-2 ctypes stub is documentation a maintainer will paste: its GemmEpilogue must be the header's
    struct, member for member (round 1 shipped a 9-field stub for a 15-field struct)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("class GemmEpilogue(C.Structure):"):doc.index("_lib.yat_gemm_epilogue_size.restype")]
    doc_fields = re.findall(r'\("(\w+)",\s*C\.(\w+)\)', block)
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = hdr[hdr.index("typedef struct yat_gemm_epilogue {"):hdr.index("} yat_gemm_epilogue;")].split("{", 1)[1]
    hdr_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ptr = "*" in decl
        base = decl.replace("const", "").replace("*", " ").split()
        ctype = "c_void_p" if ptr else {"int": "c_int", "uint32_t": "c_uint32"}[base[0]]
        for name in " ".join(base[1:]).split(","):
            hdr_fields.append((name.strip(), ctype))
    assert doc_fields == hdr_fields
    by_name = {"c_void_p": C.c_void_p, "c_int": C.c_int, "c_uint32": C.c_uint32}      # (c_uint32 is an alias of c_uint)
    assert [(n, t) for n, t in ylib.GemmEpilogue._fields_] == [(n, by_name[t]) for n, t in hdr_fields]
