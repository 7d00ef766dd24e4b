"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol include/prismer_hip.h declares.
No compute calls here (no GPU in the build container)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'prismer_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ph_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_header_symbols():
    from prismer_amd import build
    path = build.build(verbose=False)
    assert os.path.isfile(path)
    from prismer_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(_lib.lib, s), f'{s} declared in prismer_hip.h but not exported'
    assert sorted(_lib.EXPORTS) == syms, (set(_lib.EXPORTS) ^ set(syms))
    assert _lib.lib.ph_version() == _lib.ABI_VERSION == 105
    assert _lib.lib.ph_last_error() is not None


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch (error code + message, no abort)."""
    import ctypes as C
    from prismer_amd import _lib
    g = _lib.GemmArgs()
    rc = _lib.lib.ph_gemm_bf16(C.byref(g), None)
    assert rc == -1 and b'null pointer' in _lib.lib.ph_last_error()
    f = _lib.AttnFwdArgs()
    f.q = f.k = f.v = f.o = 16
    f.B = f.H = f.Sq = f.Sk = 1
    f.dh = 48
    rc = _lib.lib.ph_attention_fwd(C.byref(f), None)
    assert rc == -1 and b'head dim 48' in _lib.lib.ph_last_error()
    f.dh = 160; f.causal = 1                                                                    # HUGE resampler head dim: plain attention only
    assert _lib.lib.ph_attention_fwd(C.byref(f), None) == -1 and b'plain attention' in _lib.lib.ph_last_error()
    f.causal = 0
    f.dh = 64; f.Sk = 1 << 20; f.k_ts = 1 << 12; f.q_ts = f.v_ts = f.o_ts = 64                # K slice of 8 GiB: beyond the 32-bit tile offsets
    rc = _lib.lib.ph_attention_fwd(C.byref(f), None)
    assert rc == -1 and b'2 GiB' in _lib.lib.ph_last_error()
    # scratch sizing for hosts that own their buffers (ph_query_workspace)
    q = lambda op, *d: _lib.lib.ph_query_workspace(op, (C.c_int64 * len(d))(*d), len(d))
    assert q(0, 960, 768, 768) == 6 * 960 * 768 * 4            # GEMM split-K: 12 k-tiles -> at most 6 splits of [M][N] fp32
    assert q(0, 96, 32, 401408) == 256 * 96 * 32 * 4           # capped at 256 splits
    assert q(0, 128, 128, 64) == 0
    assert q(1, 8320, 768) == _lib.lib.ph_layernorm_bwd_blocks(8320) * 2 * 768 * 4
    assert q(2, 32, 12, 260) == 32 * 12 * 260 * 4
    assert q(3, 192) == 8 * 2 * 192 * 8
    assert q(9, 1) == -1 and q(0, 1, 2) == -1
    # output row map / generalised conv window: validated before any launch
    g = _lib.GemmArgs(); g.A = g.B = g.C = 16; g.M = g.N = g.K = 64; g.lda = g.ldb = g.ldc = 64; g.rowmap_wo = 7; g.accumulate = 1
    assert _lib.lib.ph_gemm_bf16(C.byref(g), None) == -1 and b'row map' in _lib.lib.ph_last_error()
    cnt = (C.c_int64 * 8)()
    assert _lib.lib.ph_gemm_dispatch_counts(cnt, 8, 1) == 7


def test_struct_sizes_match_header():
    """ctypes mirrors of the ABI structs must have the C layout (checked against a tiny C program)."""
    import ctypes as C
    import subprocess
    import tempfile
    from prismer_amd import _lib
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "prismer_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ph_gemm_args), sizeof(ph_layernorm_fwd_args), sizeof(ph_layernorm_bwd_args),
         sizeof(ph_attn_fwd_args), sizeof(ph_attn_bwd_args), sizeof(ph_embed_fwd_args), sizeof(ph_embed_bwd_args), sizeof(ph_rowmap));
  printf("%zu %zu %zu %zu %zu %zu %zu\n", offsetof(ph_gemm_args, split_k), offsetof(ph_layernorm_bwd_args, D), offsetof(ph_attn_bwd_args, delta),
         offsetof(ph_embed_bwd_args, dbeta), offsetof(ph_gemm_args, rowmap_add), sizeof(ph_conv_gather), sizeof(ph_conv_dgrad_item));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, 't.c'); exe = os.path.join(td, 't')
        open(src, 'w').write(prog)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), src, '-o', exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(v) for v in out[:8]]
    offs = [int(v) for v in out[8:]]
    assert sizes == [C.sizeof(_lib.GemmArgs), C.sizeof(_lib.LayerNormFwdArgs), C.sizeof(_lib.LayerNormBwdArgs), C.sizeof(_lib.AttnFwdArgs),
                     C.sizeof(_lib.AttnBwdArgs), C.sizeof(_lib.EmbedFwdArgs), C.sizeof(_lib.EmbedBwdArgs), C.sizeof(_lib.RowMap)]
    assert offs == [_lib.GemmArgs.split_k.offset, _lib.LayerNormBwdArgs.D.offset, _lib.AttnBwdArgs.delta.offset,
                    _lib.EmbedBwdArgs.dbeta.offset, _lib.GemmArgs.rowmap_add.offset, C.sizeof(_lib.ConvGather), C.sizeof(_lib.ConvDgradItem)]


def test_comm_library_exports_header_symbols():
    """libprismer_comm.so (include/prismer_comm.h): builds, loads, exports every declared entry point; argument validation
    works without a GPU and without touching RCCL."""
    import ctypes as C
    from prismer_amd import build, comm
    build.build(verbose=False)
    src = open(os.path.join(ROOT, 'include', 'prismer_comm.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    syms = sorted(set(re.findall(r'\b(ph_[a-z0-9_]+)\s*\(', src)))
    L = comm.lib()
    for s in syms:
        assert hasattr(L, s), f'{s} declared in prismer_comm.h but not exported'
    assert sorted(comm.EXPORTS) == syms
    h = C.c_void_p()
    assert L.ph_comm_init(3, 2, None, C.byref(h)) == -1 and b'bad arguments' in L.ph_comm_last_error()
    assert L.ph_allreduce_bucket(None, None, 4, 0, None) == -1
    assert L.ph_comm_world(None) == 0


def test_library_sources_emit_no_memset_or_memcpy_nodes():
    """No hipMemsetAsync / hipMemcpyAsync call in the library: inside a captured step they become memset / memcpy NODES, and small memset
    nodes replay wrongly on this ROCm (round 5).  Zeroing is done by kernels."""
    import glob
    import re
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'prismer_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(here, '*.hip')) + glob.glob(os.path.join(here, '*.h'))):
        code = re.sub(r'//[^\n]*', '', open(f).read())
        assert not re.search(r'hipMem(set|cpy)\w*Async\s*\(', code), f
