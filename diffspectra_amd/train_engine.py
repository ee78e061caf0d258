"""Training graph of the DMT on the HIP training library: forward tape + hand-written backward.

The reference trains through ``torch.autograd`` over ``models/dmt.py``; here every operation of that graph has an explicit forward
and backward kernel in ``csrc/ds_train.hip`` (C-ABI ``include/diffspectra_train.h``) over the packed-ragged layout, and this module
strings them together: ``DmtTrainGraph.forward`` records the activations a backward needs, ``backward`` walks the tape in
reverse and writes the gradient of every parameter.  PyTorch allocates the buffers, slices / concatenates them and owns the
parameters; no arithmetic of the model runs in PyTorch, and there is no CPU path (``engine.load_library`` raises without the .so).

Same de-duplicated formulation as the sampling kernels (exactly result-preserving): adaLN / time MLPs once per
molecule, ``input_lin`` split into row / column / edge parts, ``node2edge_lin`` per node, edge-side tensors once per unordered
pair (a pair row's gradient is the sum over its two directed edges - every backward operation is linear in the incoming gradient).
Dropout is the identity (stage A: the reference's p = 0 arithmetic, pinned by golden G13).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import re
from typing import Dict, List, Optional

import numpy as np
import torch

from . import engine as E

TRAIN_HEADER = os.path.join(E._ROOT, "include", "diffspectra_train.h")
ADA, ADA_STRIDE, ADA_TOP = E.ADA_COLS, E.ADA_STRIDE, E.CONSTS["DS_ADA_TOP"]
NODE_OFF, EDGE_OFF, EQUI_OFF, DIST_OFF = (E.CONSTS[k] for k in ("DS_ADA_NODE", "DS_ADA_EDGE", "DS_ADA_EQUI", "DS_ADA_DIST"))
NB = E.NB
SILU, GELU, TANH = 1, 2, 3
ADDREF = 4                 # dst_gemm `dact` code: the epilogue adds ref (a residual operand) instead of multiplying by f'(ref)


class DstGemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("a_rs", C.c_int64), ("a_cs", C.c_int64), ("B", C.c_void_p), ("b_rs", C.c_int64), ("b_cs", C.c_int64),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("bias", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("accumulate", C.c_int32), ("partial", C.c_void_p), ("partial_cap", C.c_int64), ("bf16", C.c_int32), ("_pad", C.c_int32), ("rowsum", C.c_void_p),
                ("act", C.c_int32), ("dact", C.c_int32), ("ref", C.c_void_p), ("ldref", C.c_int64), ("C2", C.c_void_p), ("ldc2", C.c_int64),
                ("drop_p", C.c_float), ("drop_stream", C.c_uint32), ("drop_seed", C.c_uint64), ("drop_ld", C.c_int64)]


class DstPiece(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("src_ld", C.c_int64), ("dst_ld", C.c_int64)]


def copy_pieces(lib, dev, dst: List[torch.Tensor], src: List[torch.Tensor], cache: dict, key: str, stream=None):
    """``dst[i].copy_(src[i])`` for many small (<= 2-D, last dimension contiguous) fp32 pieces in ONE ``dst_copy_pieces`` launch.  The device
    table is rebuilt only when a pointer changed (parameters live in the optimizer's flat buffer, the concatenated buffers are persistent)."""
    sig = tuple(t.data_ptr() for t in dst) + tuple(t.data_ptr() for t in src)
    ent = cache.get(key)
    if ent is None or ent[0] != sig:
        arr = (DstPiece * len(dst))()
        for i, (d, s_) in enumerate(zip(dst, src)):
            assert d.shape == s_.shape and d.dtype == torch.float32 and s_.dtype == torch.float32 and d.dim() <= 2
            rows, cols = (1, d.numel()) if d.dim() < 2 else (d.shape[0], d.shape[1])
            assert rows * cols < 2 ** 31                              # the kernel's index arithmetic is 32-bit
            ld = lambda t: (t.stride(0) if t.dim() == 2 and t.shape[0] > 1 else cols)
            assert (d.dim() < 2 or d.stride(-1) == 1 or cols == 1) and (s_.dim() < 2 or s_.stride(-1) == 1 or cols == 1)
            assert d.dim() >= 1 and (d.dim() == 2 or d.is_contiguous()) and (s_.dim() == 2 or s_.is_contiguous())
            arr[i] = DstPiece(src=s_.data_ptr(), dst=d.data_ptr(), rows=rows, cols=cols, src_ld=ld(s_), dst_ld=ld(d))
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        ent = (sig, raw, len(dst))
        cache[key] = ent
    E._check(lib.dst_copy_pieces(C.c_void_p(ent[1].data_ptr()), C.c_int32(ent[2]), stream if stream is not None else E._stream()), "dst_copy_pieces")


def pack_bf16_pieces(lib, dev, dst: List[torch.Tensor], src: List[torch.Tensor], cache: dict, key: str, stream=None, key_t=None):
    """``dst[i].copy_(src[i])`` with ``dst`` in bfloat16 (round to nearest even) for many small 2-D pieces in ONE ``dst_pack_bf16_pieces`` launch:
    the weights of the fused row chains, once per step.  ``key_t``: the indices of the pieces that are stored TRANSPOSED.  The device table is
    rebuilt only when a pointer changed."""
    sig = tuple(t.data_ptr() for t in dst) + tuple(t.data_ptr() for t in src)
    ent = cache.get(key)
    if ent is None or ent[0] != sig:
        arr = (DstPiece * len(dst))()
        for i, (d, s_) in enumerate(zip(dst, src)):
            assert d.dtype == torch.bfloat16 and s_.dtype == torch.float32 and d.dim() == 2 and d.stride(1) == 1 and s_.stride(1) == 1 and d.numel() < 2 ** 31
            transposed = key_t is not None and i in key_t           # dst = src^T (dst_ld < 0 in the table)
            assert tuple(d.shape) == (tuple(s_.shape)[::-1] if transposed else tuple(s_.shape))
            arr[i] = DstPiece(src=s_.data_ptr(), dst=d.data_ptr(), rows=s_.shape[0], cols=s_.shape[1], src_ld=s_.stride(0),
                              dst_ld=-d.stride(0) if transposed else d.stride(0))
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        ent = (sig, raw, len(dst))
        cache[key] = ent
    E._check(lib.dst_pack_bf16_pieces(C.c_void_p(ent[1].data_ptr()), C.c_int32(ent[2]), stream if stream is not None else E._stream()), "dst_pack_bf16_pieces")


# dst_gemm_args as one struct.pack (a ctypes Structure built field by field costs ~10 us per product, ~600 products per step)
import struct as _struct
_GEMM_STRUCT = _struct.Struct("@PqqPqqPqPiiiiPqiiPiiPqPqfIQq")
assert _GEMM_STRUCT.size == C.sizeof(DstGemmArgs), (_GEMM_STRUCT.size, C.sizeof(DstGemmArgs))
_GEMM_PACK = _GEMM_STRUCT.pack
# dst_pair_chain_args (include/diffspectra_train.h): 4 pointers, ld_feat | ada, ada_ld | 4 offsets | W3 b3 W4 b4 Wed, ld_wed | bed Wro bro |
# drop_p, stream3, stream4, pad | seed | 11 output pointers
_CHAIN_PACK = _struct.Struct("@PPP PPPPq Pq iiii PPPPP q PPP f III Q PPPPPPPPPPP").pack
# dst_pair_front_args: pos, ada, ada_ld | dist_off, shift_off, scale_off, pad | means stds e_in Wee bee Wte | X1 xs d2 e1 st en te
_FRONT_PACK = _struct.Struct("@PPP PPq iiii PPPPPP PPPPPPP").pack
# dst_node_bwd_args: 4 tables, n_tiles | dh drn ld_drn dac | f2 f1 x1 st attn | ada d_ada ada_ld | 4 offsets | WacT WnT W2T W1T | drop | seed | 5 outputs
_NODEB_PACK = _struct.Struct("@PPPPq PPqP PPPPP PPq iiii PPPP f III Q PPPPP").pack
# dst_pair_bwd_args: 4 tables, n_tiles | de dro ld_dro ded | f4 f3 xe1 st he | ada d_ada ada_ld | 4 offsets | WedT WroT W4T W3T | drop | seed | 6 outputs
_PAIRB_PACK = _struct.Struct("@PPPPq PPqP PPPPP PPq iiii PPPP f III Q PPPPPP").pack
# dst_dir_bwd_args: 4 tables, n_tiles | dc2 c0 zz st | ada d_ada ada_ld | shift_off scale_off | W2 W0T | dc0 dz part
_DIRB_PACK = _struct.Struct("@PPPPq PPPP PPq ii PP PPP").pack
# dst_node_chain_args: node_mol h_in attn | ada, ada_ld | 4 offsets | W1 b1 W2 b2 Wac Wn bn | drop_p, stream1, stream2, pad | seed | 9 outputs
_NODE_PACK = _struct.Struct("@PPP Pq iiii PPPPPPP f III Q PPPPPPPPP").pack
# dst_dir_chain_args: ac, ed, ada, ada_ld | shift_off, scale_off | W0 b0 W2 | zz st zn c0 sc0 c2
_DIR_PACK = _struct.Struct("@PPP PPPq ii PPP PPPPPP").pack


class DstLayout(C.Structure):
    _fields_ = [("B", C.c_int32), ("Nn", C.c_int32), ("Pp", C.c_int32), ("_pad", C.c_int32), ("node_off", C.c_void_p), ("pair_off", C.c_void_p)]


def train_exports() -> List[str]:
    txt = re.sub(r"/\*.*?\*/", "", open(TRAIN_HEADER).read(), flags=re.S)
    return re.findall(r"^\s*int\s+(dst_\w+)\s*\(", txt, flags=re.M)


_lib = None


def load_train_library() -> C.CDLL:
    """The training entry points live in the same shared library as the sampling path; fail loudly if any is missing."""
    global _lib
    if _lib is None:
        lib = E.load_library()
        for name in train_exports():
            getattr(lib, name).restype = C.c_int
        sizes = (C.c_int64 * 3)()
        lib.dst_struct_sizes(sizes)
        mine = [C.sizeof(DstGemmArgs), C.sizeof(DstLayout), C.sizeof(DstPiece)]
        if list(sizes) != mine:
            raise RuntimeError(f"C-ABI struct layout mismatch (training): library {list(sizes)} vs binding {mine}")
        _lib = lib
    return _lib


class MV:
    """A row-major matrix view into a device tensor: ``rows x cols`` at element offset ``off`` with row stride ``ld``."""
    __slots__ = ("t", "rows", "cols", "ld", "off")

    def __init__(self, t: torch.Tensor, rows: int, cols: int, ld: int, off: int = 0):
        self.t, self.rows, self.cols, self.ld, self.off = t, rows, cols, ld, off

    @property
    def ptr(self) -> int:
        return self.t.data_ptr() + 4 * self.off


def mv(t: torch.Tensor, c0: Optional[int] = None, c1: Optional[int] = None, r0: int = 0, r1: Optional[int] = None) -> MV:
    """View of a contiguous 2-D (or 1-D as one row) fp32 tensor, optionally restricted to columns c0:c1 / rows r0:r1.
    (Called ~1 500 times per training step: kept free of numpy and of asserts that cost more than the launch they guard.)"""
    shp = t.shape
    if len(shp) == 1:
        t2r, t2c = 1, shp[0]
    else:
        t2r = shp[0]
        t2c = shp[1] if len(shp) == 2 else (t.numel() // t2r if t2r else 0)
    if not t.is_contiguous() or t.dtype is not torch.float32:
        raise ValueError("mv() needs a contiguous fp32 tensor")
    if c0 is None:
        c0 = 0
    if c1 is None:
        c1 = t2c
    if r1 is None:
        r1 = t2r
    return MV(t, r1 - r0, c1 - c0, t2c, r0 * t2c + c0)


class Ops:
    """ctypes wrappers over the dst_* entry points, issued on torch's current stream."""

    def __init__(self, device):
        self.lib = load_train_library()
        self.dev = torch.device(device)
        self.scratch = torch.empty(48 * 1024 * 1024, dtype=torch.float32, device=self.dev)     # split-K partials / column sums
        self.bf16 = False      # config.training.precision == 'bf16': every GEMM rounds its operands to bf16 (fp32 accumulate, fp32 storage)
        # Weight-gradient products on a SIDE stream (DmtTrainGraph.backward switches it on): nothing downstream of a backward pass reads
        # dW, so the ~200 split-K products of a step need not sit in the dependent chain of input-gradient kernels - they fill the CUs the
        # small kernels of that chain leave idle.  The side stream has its own split-K scratch; operands are kept alive until join_dw().
        self.async_dw = False
        self.stream_ptr = None
        self._side = None
        self._side_scratch = None
        self._sides = []           # the weight-gradient streams (DIFFSPECTRA_DW_STREAMS of them, round-robin), each with its split-K scratch
        self._side_of = {}         # output pointer -> stream index: products that accumulate into the same gradient stay on one stream, in order
        self._side_next = 0
        self._dw_keep = []

    def _s(self):
        """The HIP stream the kernels are issued on.  ``torch.cuda.current_stream()`` costs ~8 us and a step makes ~1 200 calls: the training
        entry points fetch it once (``begin``) and every launch of the step reuses the handle (``end`` drops it)."""
        return self.stream_ptr if self.stream_ptr is not None else E._stream()

    def begin(self):
        self.main_stream = torch.cuda.current_stream(self.dev)
        self.stream_ptr = C.c_void_p(self.main_stream.cuda_stream)

    def end(self):
        self.stream_ptr = None
        self.main_stream = None
        self.cur_stream = None

    def gemm(self, A: MV, Bm: MV, Cm: MV, ta: bool, tb: bool, bias: Optional[torch.Tensor] = None, acc: bool = False,
             rowsum: Optional[torch.Tensor] = None, act: int = 0, dact: int = 0, ref: Optional[MV] = None, out2: Optional[MV] = None,
             drop=None):
        """``drop = (p, seed, stream_id, row_length)``: the Philox dropout mask of element (m, n) is taken at index m * row_length + n."""
        M, K = (A.cols, A.rows) if ta else (A.rows, A.cols)
        a_rs, a_cs = (1, A.ld) if ta else (A.ld, 1)
        K2, N = (Bm.cols, Bm.rows) if tb else (Bm.rows, Bm.cols)
        b_rs, b_cs = (1, Bm.ld) if tb else (Bm.ld, 1)
        assert K == K2 and Cm.rows == M and Cm.cols == N, (M, K, K2, N, Cm.rows, Cm.cols)
        if bias is not None:
            assert bias.numel() == N
        if rowsum is not None:
            assert rowsum.numel() == M and rowsum.is_contiguous()
        dp = drop[0] if drop and drop[0] > 0 else 0.0
        args = _GEMM_PACK(A.ptr, a_rs, a_cs, Bm.ptr, b_rs, b_cs, Cm.ptr, Cm.ld, 0 if bias is None else bias.data_ptr(), M, N, K, int(acc),
                          self.scratch.data_ptr(), self.scratch.numel(), int(self.bf16), 0, 0 if rowsum is None else rowsum.data_ptr(), act, dact,
                          0 if ref is None else ref.ptr, 0 if ref is None else ref.ld, 0 if out2 is None else out2.ptr, 0 if out2 is None else out2.ld,
                          float(dp), int(drop[2]) if drop else 0, int(drop[1]) if drop else 0, int(drop[3]) if drop else 0)
        if ref is not None:
            assert ref.rows == M and ref.cols == N
        if out2 is not None:
            assert out2.rows == M and out2.cols == N
        E._check(self.lib.dst_gemm(args, self._s()), "dst_gemm")

    def colsum(self, X: MV, out: torch.Tensor, acc: bool = False, param_grad: bool = False):
        """``param_grad``: the sums are a parameter gradient (nothing downstream of the pass reads them) - with the weight-gradient stream on
        they go there, like ``lin_bwd_w``; X must then not be overwritten before ``join_dw``."""
        assert out.numel() == X.cols
        if param_grad and self.async_dw:
            self._to_side(X.t, out, out=out)
            saved = (self.scratch, self.stream_ptr)
            self.scratch, self.stream_ptr = self._side_scratch, C.c_void_p(self._side.cuda_stream)
            try:
                return self.colsum(X, out, acc)
            finally:
                self.scratch, self.stream_ptr = saved
        E._check(self.lib.dst_colsum(C.c_void_p(X.ptr), C.c_int64(X.ld), C.c_int32(X.rows), C.c_int32(X.cols), E._ptr(out), C.c_int32(int(acc)),
                                     E._ptr(self.scratch), C.c_int64(self.scratch.numel()), self._s()), "dst_colsum")

    def _to_side(self, *keep, out=None):
        """Pick the weight-gradient stream of this product (``out``: its output tensor - the same output always goes to the same stream) and let
        it wait for the stream the operands were produced on (main, or the node stream inside a node section).  Several streams: a weight
        gradient is a split-K product plus its reduction, two dependent launches that fill a fraction of the chip - on one stream they ran one
        after the other and that stream, not the main one, ended the backward (leaving every weight gradient out shortened the step by 3.9 ms)."""
        if not self._sides:
            n = max(1, int(os.environ.get("DIFFSPECTRA_DW_STREAMS", "2")))     # (2 -> 3: no further gain; profiles/r05_train_dw_streams.txt)
            self._sides = [(torch.cuda.Stream(device=self.dev), torch.empty_like(self.scratch)) for _ in range(n)]
        key = None if out is None else out.data_ptr()
        k = self._side_of.get(key) if key is not None else None
        if k is None:
            k = self._side_next
            self._side_next = (k + 1) % len(self._sides)
            if key is not None:
                self._side_of[key] = k
        self._side, self._side_scratch = self._sides[k]
        src = getattr(self, "cur_stream", None)
        if src is None:
            src = self.main_stream if getattr(self, "main_stream", None) is not None else torch.cuda.current_stream(self.dev)
        self._side.wait_stream(src)
        self._dw_keep.append(keep)

    # y = x W^T + b ; dx (+)= dy W ; dW = dy^T x ; db = colsum(dy)
    def lin_fwd(self, x: MV, W: MV, b, y: MV, act: int = 0, out2: Optional[MV] = None, drop=None):
        """``act`` + ``out2``: y keeps the pre-activation (the backward's reference), out2 = drop(f(y)); ``act`` alone: y = drop(f(.))."""
        self.gemm(x, W, y, False, True, bias=b, act=act, out2=out2, drop=drop)

    def lin_bwd_x(self, dy: MV, W: MV, dx: MV, acc: bool = False, dact: int = 0, ref: Optional[MV] = None, drop=None):
        """``dact`` + ``ref``: dx = (dy W) * f'(ref) (* the dropout mask ``drop`` of the activated tensor): the gradient in front of
        ``drop(f(.))`` in one product."""
        self.gemm(dy, W, dx, False, False, acc=acc, dact=dact, ref=ref, drop=drop)

    def lin_bwd_w(self, dy: MV, x: MV, dW: MV, db: Optional[torch.Tensor] = None, acc: bool = False):
        if not self.async_dw:
            self.gemm(dy, x, dW, True, False, acc=acc, rowsum=db)       # db = column sums of dy = row sums of dy^T, fused into the product
            return
        self._to_side(dy.t, x.t, dW.t, db, out=dW.t)                    # dy (and x) are complete on their stream at this point
        main_scratch, self.scratch = self.scratch, self._side_scratch
        main_ptr, self.stream_ptr = self.stream_ptr, C.c_void_p(self._side.cuda_stream)
        try:
            self.gemm(dy, x, dW, True, False, acc=acc, rowsum=db)
        finally:
            self.scratch, self.stream_ptr = main_scratch, main_ptr

    # ---- node stream (forward): the node-row chain of a block (4 600 rows: ten launches, each shorter than its launch latency) runs beside
    #      the pair-row chain instead of in front of it.  Rules that make it safe with torch's caching allocator (all tensors come from
    #      the MAIN stream's pool): every node section starts by waiting for everything the main stream has been given so far (whatever
    #      memory the section allocates was freed before that point, so its earlier users are covered), and nothing a section touches is
    #      freed before the join at the end of the pass (the graph holds the references).
    def node_section(self):
        return _NodeSection(self)

    def node_event(self):
        ev = torch.cuda.Event()
        ev.record(self._node)
        return ev

    def main_wait(self, ev=None):
        """The main stream waits for the node stream (as of now) or for one recorded event."""
        if ev is None:
            self.main_stream.wait_stream(self._node)
        else:
            self.main_stream.wait_event(ev)

    def join_dw(self):
        """The main stream waits for every weight-gradient product issued so far; their operands may be reused after it."""
        if self._sides and self._dw_keep:
            main = self.main_stream if getattr(self, "main_stream", None) is not None else torch.cuda.current_stream(self.dev)
            for st, _ in self._sides:
                main.wait_stream(st)
        self._side_of = {}
        self._dw_keep = []

    def act_fwd(self, x, y, kind):
        E._check(self.lib.dst_act_fwd(E._ptr(x), E._ptr(y), C.c_int64(x.numel()), C.c_int32(kind), self._s()), "dst_act_fwd")

    def act_bwd(self, dy, ref, dx, kind):
        E._check(self.lib.dst_act_bwd(E._ptr(dy), E._ptr(ref), E._ptr(dx), C.c_int64(dy.numel()), C.c_int32(kind), self._s()), "dst_act_bwd")

    def dropout(self, x, p, seed, stream_id):
        """In place; the same (seed, stream_id) on the gradient is the backward."""
        if p > 0.0:
            E._check(self.lib.dst_dropout(E._ptr(x), E._ptr(x), C.c_int64(x.numel()), C.c_float(p), C.c_uint64(seed), C.c_uint32(stream_id), self._s()),
                     "dst_dropout")

    def axpy(self, a, x, y):
        E._check(self.lib.dst_axpy(C.c_float(a), E._ptr(x), E._ptr(y), C.c_int64(x.numel()), self._s()), "dst_axpy")

    def lnmod_fwd(self, x, Cc, seg, mul, B, ada, sh, sc, y, stats):
        E._check(self.lib.dst_lnmod_fwd(E._ptr(x), C.c_int32(Cc), E._ptr(seg), C.c_int32(mul), C.c_int32(B), E._ptr(ada), C.c_int64(ADA), C.c_int32(sh),
                                        C.c_int32(sc), E._ptr(y), E._ptr(stats), self._s()), "dst_lnmod_fwd")

    def lnmod_bwd(self, dy, x, stats, Cc, seg, mul, B, ada, d_ada, sh, sc, dx, acc):
        E._check(self.lib.dst_lnmod_bwd(E._ptr(dy), E._ptr(x), E._ptr(stats), C.c_int32(Cc), E._ptr(seg), C.c_int32(mul), C.c_int32(B), E._ptr(ada),
                                        E._ptr(d_ada), C.c_int64(ADA), C.c_int32(sh), C.c_int32(sc), E._ptr(dx), C.c_int32(int(acc)), E._ptr(self.scratch),
                                        C.c_int64(self.scratch.numel()), self._s()), "dst_lnmod_bwd")

    def gate_add_fwd(self, r, z, Cc, seg, mul, B, ada, g, out):
        E._check(self.lib.dst_gate_add_fwd(E._ptr(r), E._ptr(z), C.c_int32(Cc), E._ptr(seg), C.c_int32(mul), C.c_int32(B), E._ptr(ada), C.c_int64(ADA),
                                           C.c_int32(g), E._ptr(out), self._s()), "dst_gate_add_fwd")

    def pair_chain_fwd(self, TL, u, n2e_bias, e_in, feat, ld_feat, ada, g1, sh, sc, g2, W3, b3, W4, b4, Wed, ld_wed, bed, Wro, bro, drop, out):
        """The pair rows of a block behind the attention as one kernel (``dst_pair_chain_fwd``, bf16 products).  ``drop = (p, seed, stream3,
        stream4)``; ``out``: dict with e_out, ed, ro and - when the tape is kept - he, xe1, st, ye1, f3, s3, f4, X2."""
        ptr = lambda t: 0 if t is None else t.data_ptr()
        assert all(w_.dtype == torch.bfloat16 for w_ in (W3, W4, Wed, Wro))          # (dst_pack_bf16_pieces / .to(torch.bfloat16): nearest even)
        args = _CHAIN_PACK(*TL.pair_tables, ptr(u), ptr(n2e_bias), ptr(e_in), ptr(feat), ld_feat, ptr(ada), ADA, g1, sh, sc, g2, ptr(W3), ptr(b3), ptr(W4), ptr(b4),
                           ptr(Wed), ld_wed, ptr(bed), ptr(Wro), ptr(bro), float(drop[0]), int(drop[2]), int(drop[3]), 0, int(drop[1]),
                           *(ptr(out.get(k)) for k in ("he", "xe1", "st", "ye1", "f3", "s3", "f4", "e_out", "X2", "ed", "ro")))
        E._check(self.lib.dst_pair_chain_fwd(C.byref(TL.c), args, self._s()), "dst_pair_chain_fwd")

    def node_chain_fwd(self, TL, h_in, attn, ada, g1, sh, sc, g2, W1, b1, W2, b2, Wac, Wn, bn, drop, out):
        """The node rows of a block behind the attention as one kernel (``dst_node_chain_fwd``, bf16 products).  ``drop = (p, seed, stream1,
        stream2)``; ``out``: dict with h_out, ac, rn and - when the tape is kept - x1, st, y1, f1, s1, f2."""
        ptr = lambda t: 0 if t is None else t.data_ptr()
        assert all(w_.dtype == torch.bfloat16 for w_ in (W1, W2, Wac, Wn))
        args = _NODE_PACK(TL.node_mol_ptr, ptr(h_in), ptr(attn), ptr(ada), ADA, g1, sh, sc, g2, ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(Wac), ptr(Wn), ptr(bn),
                          float(drop[0]), int(drop[2]), int(drop[3]), 0, int(drop[1]),
                          *(ptr(out.get(k)) for k in ("x1", "st", "y1", "f1", "s1", "f2", "h_out", "ac", "rn")))
        E._check(self.lib.dst_node_chain_fwd(C.byref(TL.c), args, self._s()), "dst_node_chain_fwd")

    def dir_chain_fwd(self, TL, ac, ed, ada, sh, sc, W0, b0, W2, out):
        """The directed rows of a block as one kernel (``dst_dir_chain_fwd``, bf16 products).  ``out``: dict with c2 and - when the tape is kept -
        zz, st, zn, c0, sc0."""
        ptr = lambda t: 0 if t is None else t.data_ptr()
        assert W0.dtype == torch.bfloat16 and W2.dtype == torch.bfloat16
        args = _DIR_PACK(*TL.pair_tables, ptr(ac), ptr(ed), ptr(ada), ADA, sh, sc, ptr(W0), ptr(b0), ptr(W2), *(ptr(out.get(k)) for k in ("zz", "st", "zn", "c0", "sc0", "c2")))
        E._check(self.lib.dst_dir_chain_fwd(C.byref(TL.c), args, self._s()), "dst_dir_chain_fwd")

    def pair_chain_bwd(self, TL, de, dro, ld_dro, ded, f4, f3, xe1, st, he, ada, d_ada, g1, sh, sc, g2, WedT, WroT, W4T, W3T, drop, dfeat, df4, df3, de_in, dhe):
        """Backward of the pair rows of a block behind the attention as one kernel + its finishing kernel (``dst_pair_chain_bwd``).  ``dro``: a
        data pointer (the read-out slice's gradient is a column window of a wider tensor) with row stride ``ld_dro``; ``drop = (p, seed, stream3,
        stream4)``."""
        assert all(w_.dtype == torch.bfloat16 for w_ in (WedT, WroT, W4T, W3T))
        tt = TL.pair_tiles
        need = tt[4] * 256
        if getattr(self, "_pairb_part", None) is None or self._pairb_part.numel() < need:
            self._pairb_part = torch.empty(max(need, 1), dtype=torch.float32, device=self.dev)
        dp_ = lambda t: t.data_ptr()
        args = _PAIRB_PACK(tt[0], tt[1], tt[2], tt[3], tt[4], dp_(de), int(dro), ld_dro, dp_(ded), dp_(f4), dp_(f3), dp_(xe1), dp_(st), dp_(he), dp_(ada), dp_(d_ada), ADA,
                           g1, sh, sc, g2, dp_(WedT), dp_(WroT), dp_(W4T), dp_(W3T), float(drop[0]), int(drop[2]), int(drop[3]), 0, int(drop[1]),
                           dp_(dfeat), dp_(df4), dp_(df3), dp_(de_in), dp_(dhe), self._pairb_part.data_ptr())
        E._check(self.lib.dst_pair_chain_bwd(C.byref(TL.c), args, self._s()), "dst_pair_chain_bwd")

    def node_chain_bwd(self, TL, dh, drn, ld_drn, dac, f2, f1, x1, st, attn, ada, d_ada, g1, sh, sc, g2, WacT, WnT, W2T, W1T, drop, df2, df1, dh_in, dattn):
        """Backward of the node rows of a block behind the attention as one kernel + its finishing kernel (``dst_node_chain_bwd``).  ``drn``: a data
        pointer with row stride ``ld_drn``; ``drop = (p, seed, stream1, stream2)``."""
        assert all(w_.dtype == torch.bfloat16 for w_ in (WacT, WnT, W2T, W1T))
        tt = TL.node_tiles
        need = tt[4] * 1024
        part = getattr(self, "_nodeb_part", None)
        if part is None or part.numel() < need:
            part = self._nodeb_part = torch.empty(max(need, 1), dtype=torch.float32, device=self.dev)
        dp_ = lambda t: t.data_ptr()
        args = _NODEB_PACK(tt[0], tt[1], tt[2], tt[3], tt[4], dp_(dh), int(drn), ld_drn, dp_(dac), dp_(f2), dp_(f1), dp_(x1), dp_(st), dp_(attn), dp_(ada), dp_(d_ada), ADA,
                           g1, sh, sc, g2, dp_(WacT), dp_(WnT), dp_(W2T), dp_(W1T), float(drop[0]), int(drop[2]), int(drop[3]), 0, int(drop[1]),
                           dp_(df2), dp_(df1), dp_(dh_in), dp_(dattn), part.data_ptr())
        E._check(self.lib.dst_node_chain_bwd(C.byref(TL.c), args, self._s()), "dst_node_chain_bwd")

    def dir_chain_bwd(self, TL, dc2, c0, zz, st, ada, d_ada, sh, sc, W2, W0T, dc0, dz):
        """Backward of the directed rows of a block as one kernel + its finishing kernel (``dst_dir_chain_bwd``)."""
        assert W0T.dtype == torch.bfloat16 and W2.dtype == torch.float32
        tt = TL.dir_tiles
        need = tt[4] * 512
        if getattr(self, "_dirb_part", None) is None or self._dirb_part.numel() < need:
            self._dirb_part = torch.empty(max(need, 1), dtype=torch.float32, device=self.dev)
        args = _DIRB_PACK(tt[0], tt[1], tt[2], tt[3], tt[4], dc2.data_ptr(), c0.data_ptr(), zz.data_ptr(), st.data_ptr(), ada.data_ptr(), d_ada.data_ptr(), ADA, sh, sc,
                          W2.data_ptr(), W0T.data_ptr(), dc0.data_ptr(), dz.data_ptr(), self._dirb_part.data_ptr())
        E._check(self.lib.dst_dir_chain_bwd(C.byref(TL.c), args, self._s()), "dst_dir_chain_bwd")

    def pair_front_fwd(self, TL, pos, ada, dist_off, sh, sc, means, stds, e_in, Wee, bee, Wte, out):
        """The pair rows of a block in front of the attention as one kernel (``dst_pair_front_fwd``, bf16 products).  ``out``: dict with X1, te and -
        when the tape is kept - xs, d2, e1, st, en."""
        ptr = lambda t: 0 if t is None else t.data_ptr()
        assert Wee.dtype == torch.bfloat16 and Wte.dtype == torch.bfloat16
        args = _FRONT_PACK(*TL.pair_tables, ptr(pos), ptr(ada), ADA, dist_off, sh, sc, 0, ptr(means), ptr(stds), ptr(e_in), ptr(Wee), ptr(bee), ptr(Wte),
                           *(ptr(out.get(k)) for k in ("X1", "xs", "d2", "e1", "st", "en", "te")))
        E._check(self.lib.dst_pair_front_fwd(C.byref(TL.c), args, self._s()), "dst_pair_front_fwd")

    def gate_add_bwd(self, dout, z, Cc, seg, mul, B, ada, d_ada, g, dr, acc_r, dz, drop=None):
        """``drop = (p, seed, stream_id)``: ``z`` was a dropout's output; ``dz`` is then the gradient in FRONT of that dropout."""
        p_, seed, stream = drop if drop and drop[0] > 0 else (0.0, 0, 0)
        E._check(self.lib.dst_gate_add_bwd(E._ptr(dout), E._ptr(z), C.c_int32(Cc), E._ptr(seg), C.c_int32(mul), C.c_int32(B), E._ptr(ada), E._ptr(d_ada),
                                           C.c_int64(ADA), C.c_int32(g), E._ptr(dr), C.c_int32(int(acc_r)), E._ptr(dz), C.c_float(p_), C.c_uint64(seed),
                                           C.c_uint32(stream), self._s()), "dst_gate_add_bwd")


class _NodeSection:
    def __init__(self, ops):
        self.o = ops

    def __enter__(self):
        o = self.o
        if getattr(o, "_node", None) is None:
            o._node = torch.cuda.Stream(device=o.dev)
            o._node_ptr = C.c_void_p(o._node.cuda_stream)
            o._node_scratch = torch.empty(4 * 1024 * 1024, dtype=torch.float32, device=o.dev)
        o._node.wait_stream(o.main_stream)
        self.saved = (o.stream_ptr, o.scratch, getattr(o, "cur_stream", None))
        o.stream_ptr, o.scratch, o.cur_stream = o._node_ptr, o._node_scratch, o._node
        return self

    def __exit__(self, *exc):
        self.o.stream_ptr, self.o.scratch, self.o.cur_stream = self.saved
        return False


class TrainLayout:
    """The packed-ragged tables of ``engine.Layout`` plus what the training kernels need (dense <-> packed index tensors)."""

    def __init__(self, node_mask: torch.Tensor, device):
        self.L = E.Layout(node_mask, device)
        L = self.L
        self.B, self.N, self.Nn, self.Pp = L.B, L.N, L.Nn, L.Pp
        self.c = DstLayout(B=L.B, Nn=L.Nn, Pp=L.Pp, _pad=0, node_off=L.t["node_off"].data_ptr(), pair_off=L.t["pair_off"].data_ptr())
        nd = L.t["node_dense"].long()
        a, b = L.t["pair_a"].long(), L.t["pair_b"].long()
        self.node_dense = nd                                              # [Nn] -> row of the dense [B*N] node arrays
        self.pair_dense = nd[a] * L.N + (nd[b] % L.N)                     # [Pp] -> row (b, i, j) of the dense [B*N*N] edge arrays, i < j
        self.pair_dense_t = nd[b] * L.N + (nd[a] % L.N)                   # the transposed cell (b, j, i)
        self.node_mol = L.t["node_mol"].long()
        self.pair_mol = L.t["pair_mol"].long()
        self.node_off, self.pair_off = L.t["node_off"], L.t["pair_off"]
        # device tables of the flat-tile kernels (dst_pair_*_fwd, dst_dir_chain_fwd): node rows of a pair's atoms, its molecule
        self.pair_tables = (L.t["pair_a"].data_ptr(), L.t["pair_b"].data_ptr(), L.t["pair_mol"].data_ptr())
        self.node_mol_ptr = L.t["node_mol"].data_ptr()                     # [Nn] int32: molecule of a node row (dst_node_chain_fwd)
        # molecule-aligned 32-row tiles of the pair rows, the DIRECTED rows and the node rows (dst_pair_chain_bwd / dst_dir_chain_bwd /
        # dst_node_chain_bwd: their adaLN sums are per molecule): (first row, row count, molecule) per tile, first tile per molecule, number of tiles
        po = L.t["pair_off"].cpu().numpy().astype(np.int64)
        i32 = lambda v: torch.tensor(v if len(v) else [0], dtype=torch.int32, device=device)
        self._tile_tensors = []
        no = L.t["node_off"].cpu().numpy().astype(np.int64)
        for mul, name in ((1, "pair_tiles"), (2, "dir_tiles"), (0, "node_tiles")):
            row0, rows, mol, off = [], [], [], [0]
            for m in range(L.B):
                nr, r0 = (mul * int(po[m + 1] - po[m]), mul * int(po[m])) if mul else (int(no[m + 1] - no[m]), int(no[m]))
                for k in range(0, nr, 32):
                    row0.append(r0 + k); rows.append(min(32, nr - k)); mol.append(m)
                off.append(len(row0))
            tens = (i32(row0), i32(rows), i32(mol), i32(off))
            self._tile_tensors.append(tens)
            setattr(self, name, tuple(t.data_ptr() for t in tens) + (len(row0),))

    def pack_nodes(self, dense: torch.Tensor) -> torch.Tensor:
        return dense.reshape(self.B * self.N, -1).index_select(0, self.node_dense).contiguous()

    def pack_pairs(self, dense: torch.Tensor) -> torch.Tensor:
        return dense.reshape(self.B * self.N * self.N, -1).index_select(0, self.pair_dense).contiguous()

    def unpack_nodes(self, packed: torch.Tensor) -> torch.Tensor:
        out = torch.zeros(self.B * self.N, packed.shape[1], dtype=packed.dtype, device=packed.device)
        out[self.node_dense] = packed
        return out.reshape(self.B, self.N, -1)

    def unpack_pairs(self, packed: torch.Tensor) -> torch.Tensor:
        out = torch.zeros(self.B * self.N * self.N, packed.shape[1], dtype=packed.dtype, device=packed.device)
        out[self.pair_dense] = packed
        out[self.pair_dense_t] = packed
        return out.reshape(self.B, self.N, self.N, -1)


ADA_PARTS = (("node_time_mlp.1", NODE_OFF, 1536), ("edge_time_mlp.1", EDGE_OFF, 384), ("equi_update.time_mlp.1", EQUI_OFF, 512),
             ("dist_layer.time_mlp.1", DIST_OFF, 2))


class DmtTrainGraph:
    """Forward tape and backward of one DMT evaluation (conditioning embedding given) on packed tensors.

    ``params``: name -> fp32 device tensor (reference names, no ``module.`` prefix).  ``grads`` (same names) receives the gradients."""

    def __init__(self, params: Dict[str, torch.Tensor], config, device):
        self.p = params
        self.cfg = config
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("the training graph runs on an MI355X only; diffspectra_amd has no CPU path")
        self.ops = Ops(self.dev)
        self.lib = self.ops.lib
        self.edge_th = float(config.model.edge_quan_th)
        self.cutoff = float(config.model.spatial_cut_off)

    # FF dropout of the training forward (dmt.py:114-120): probability and the seed of this evaluation's Philox streams; 0 = identity
    dropout_p = 0.0
    dropout_seed = 0

    # ------------------------------------------------------------------ helpers
    def f(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.dev)

    def z(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.dev)

    # ---- concatenated weights: Linears that read the same input are evaluated as ONE product (q | k | v; lin_edge0 | lin_edge1; the row | col
    #      parts of input_lin; every *time_mlp as the adaLN table) from per-step copies of the parameters in one buffer each.  The copies are
    #      two multi-tensor launches per step (torch._foreach_copy_), the gradients of the concatenated buffers are scattered back the same way.
    def cat_plan(self):
        """[(buffer name, shape)], [(buffer name, index expression, parameter name)] - where every parameter piece lives in its buffer."""
        bufs = [("Wada", (ADA, 1024)), ("bada", (ADA,)), ("Wqkv", (NB, 768, 256)), ("bqkv", (NB, 768)), ("Wte", (NB, 512, 64)), ("Wac", (NB, 512, 256))]
        pieces = []
        for blk in range(NB):
            bp = f"e_block_{blk}."
            for name, off, rows in ADA_PARTS:
                o = blk * ADA_STRIDE + off
                pieces.append(("Wada", (slice(o, o + rows),), bp + name + ".weight"))
                pieces.append(("bada", (slice(o, o + rows),), bp + name + ".bias"))
            for k, nm in enumerate(("lin_query", "lin_key", "lin_value")):
                rows = 252 if k < 2 else 256
                pieces.append(("Wqkv", (blk, slice(256 * k, 256 * k + rows)), bp + f"attn_mpnn.{nm}.weight"))
                pieces.append(("bqkv", (blk, slice(256 * k, 256 * k + rows)), bp + f"attn_mpnn.{nm}.bias"))
            pieces.append(("Wte", (blk, slice(0, 252)), bp + "attn_mpnn.lin_edge0.weight"))
            pieces.append(("Wte", (blk, slice(256, 512)), bp + "attn_mpnn.lin_edge1.weight"))
            pieces.append(("Wac", (blk, slice(0, 256)), (bp + "equi_update.input_lin.weight", slice(0, 256))))      # h_row part (columns 0..255)
            pieces.append(("Wac", (blk, slice(256, 512)), (bp + "equi_update.input_lin.weight", slice(256, 512))))  # h_col part
        pieces.append(("Wada", (slice(ADA_TOP, ADA_TOP + 2),), "dist_layer.time_mlp.1.weight"))
        pieces.append(("bada", (slice(ADA_TOP, ADA_TOP + 2),), "dist_layer.time_mlp.1.bias"))
        return bufs, pieces

    def _piece_views(self, bufs, tensors):
        """(views into the concatenated buffers, the matching parameter(-shaped) tensors) in plan order."""
        _, pieces = self.cat_plan()
        dst, src = [], []
        for bname, idx, pname in pieces:
            dst.append(bufs[bname][idx])
            if isinstance(pname, tuple):
                src.append(tensors[pname[0]][:, pname[1]])
            else:
                src.append(tensors[pname])
        return dst, src

    def prepare_weights(self, cache: Optional[dict] = None):
        """Fill the concatenated weight buffers from the current parameters (once per step: both forwards of a step share them)."""
        if cache is None:
            cache = {}
        if "bufs" not in cache:
            shapes, _ = self.cat_plan()
            cache["bufs"] = {n: self.z(*shape) for n, shape in shapes}          # padding rows (252..255 of q / k / lin_edge0) stay zero
            cache["grads"] = {n: self.z(*shape) for n, shape in shapes}
        dst, src = self._piece_views(cache["bufs"], self.p)
        copy_pieces(self.lib, self.dev, dst, src, cache, "table_fwd", self.ops._s())
        self.cat, self.dcat, self._cat_cache = cache["bufs"], cache["grads"], cache
        if self.ops.bf16:
            # the fused row chains take their weights as bf16 (csrc/ds_train_chain.hip: the per-tile weight stream from L2 bounds them)
            if "wb" not in cache:
                shapes = dict(W3=(128, 64), W4=(64, 128), Wed=(256, 128), Wro=(16, 64), Wee=(64, 128), Wte=(512, 64), W0=(256, 256), W2=(3, 256),
                              F1=(512, 256), F2=(256, 512), Wac=(512, 256), Wn=(64, 256), W0T=(256, 256),
                              WedT=(128, 256), WroT=(64, 16), W4T=(128, 64), W3T=(64, 128), WacT=(256, 512), WnT=(256, 64), F2T=(512, 256), F1T=(256, 512))
                cache["wb"] = {n: torch.zeros(NB, *sh, dtype=torch.bfloat16, device=self.dev) for n, sh in shapes.items()}
            wb, p = cache["wb"], self.p
            dst, src, tset = [], [], set()
            for i in range(NB):
                bp = f"e_block_{i}."
                for n, t in (("W3", p[bp + "ff_linear3.weight"]), ("W4", p[bp + "ff_linear4.weight"]), ("Wed", p[bp + "equi_update.input_lin.weight"][:, 512:640]),
                             ("Wro", p[f"edge_{i}.weight"]), ("Wee", p[bp + "edge_emb.weight"]), ("Wte", cache["bufs"]["Wte"][i]),
                             ("W0", p[bp + "equi_update.coord_mlp.0.weight"]), ("W2", p[bp + "equi_update.coord_mlp.2.weight"]),
                             ("F1", p[bp + "ff_linear1.weight"]), ("F2", p[bp + "ff_linear2.weight"]), ("Wac", cache["bufs"]["Wac"][i]),
                             ("Wn", p[f"node_{i}.weight"])):
                    dst.append(wb[n][i])
                    src.append(t)
                # transposed copies ([in][out]): the B operands of the input-gradient products of the fused backward kernels
                for n, t in (("W0T", p[bp + "equi_update.coord_mlp.0.weight"]), ("WedT", p[bp + "equi_update.input_lin.weight"][:, 512:640]),
                             ("WroT", p[f"edge_{i}.weight"]), ("W4T", p[bp + "ff_linear4.weight"]), ("W3T", p[bp + "ff_linear3.weight"]),
                             ("WacT", cache["bufs"]["Wac"][i]), ("WnT", p[f"node_{i}.weight"]), ("F2T", p[bp + "ff_linear2.weight"]),
                             ("F1T", p[bp + "ff_linear1.weight"])):
                    tset.add(len(dst))
                    dst.append(wb[n][i])
                    src.append(t)
            pack_bf16_pieces(self.lib, self.dev, dst, src, cache, "table_bf16", self.ops._s(), key_t=tset)
            self.wb = wb
        return cache

    def scatter_cat_grads(self, gw):
        """Gradients of the concatenated buffers -> the parameters' gradient buffers (input_lin's edge part and bias are written directly)."""
        _, pieces = self.cat_plan()
        tgt = {}
        for _, _, pname in pieces:
            name = pname[0] if isinstance(pname, tuple) else pname
            if name not in tgt:
                tgt[name] = gw(name) if name not in self._gw_done else self._gw_done[name]
        src, dst = self._piece_views(self.dcat, tgt)
        copy_pieces(self.lib, self.dev, dst, src, self._cat_cache, "table_bwd", self.ops._s())

    def _geom_fwd(self, TL, pos, ada, dist_off, prefix, X, ldx, col0, xs, d2s):
        E._check(self.lib.dst_geom_fwd(C.byref(TL.c), E._ptr(pos), E._ptr(ada), C.c_int64(ADA), C.c_int32(dist_off),
                                       E._ptr(self.p[prefix + "means.weight"]), E._ptr(self.p[prefix + "stds.weight"]),
                                       C.c_void_p(X.data_ptr() + 4 * col0), C.c_int64(ldx), E._ptr(xs), E._ptr(d2s), self.ops._s()), "dst_geom_fwd")

    def _geom_bwd(self, TL, pos, ada, d_ada, dist_off, prefix, xs, d2s, g1, g2, dms, dd2, dpos):
        E._check(self.lib.dst_geom_bwd(C.byref(TL.c), E._ptr(pos), E._ptr(ada), E._ptr(d_ada), C.c_int64(ADA), C.c_int32(dist_off),
                                       E._ptr(self.p[prefix + "means.weight"]), E._ptr(self.p[prefix + "stds.weight"]), E._ptr(xs), E._ptr(d2s),
                                       E._ptr(g1), C.c_int64(g1.shape[1]), E._ptr(g2), C.c_int64(0 if g2 is None else g2.shape[1]), E._ptr(dms),
                                       E._ptr(dd2), E._ptr(dpos), self.ops._s()), "dst_geom_bwd")

    # ------------------------------------------------------------------ forward
    def forward(self, TL: TrainLayout, xn, ex, noise_level, ctx_emb, cond_n=None, cond_e=None, save: bool = True):
        """``xn [Nn,9]`` / ``ex [Pp,2]`` packed noisy state, ``noise_level [B]``, ``ctx_emb [B,1024]`` = cond_lin(SpecFormer(context)),
        ``cond_n`` / ``cond_e`` packed self-conditioning prediction or None (dmt.py:332-345).  Returns (pos [Nn,3], atom_pred [Nn,6],
        edge_pred [Pp,2]) and keeps the tape in ``self.t`` when ``save``."""
        o, p, lib = self.ops, self.p, self.lib
        B, Nn, Pp = TL.B, TL.Nn, TL.Pp
        D = 2 * Pp
        t: Dict[str, object] = dict(TL=TL, first=cond_n is None, drop=(self.dropout_p, self.dropout_seed))
        dp, dseed = self.dropout_p, self.dropout_seed
        s = self.ops._s
        # ---- time embedding + adaLN table (dmt.py:249-257,353-357; every *time_mlp)
        tf = self.f(B, 17)
        E._check(lib.dst_time_feat_fwd(E._ptr(noise_level), E._ptr(p["time_mlp.0.weights"]), C.c_int32(B), E._ptr(tf), s()), "dst_time_feat_fwd")
        tm1, tg, temb, st = self.f(B, 1024), self.f(B, 1024), self.f(B, 1024), self.f(B, 1024)
        o.lin_fwd(mv(tf), mv(p["time_mlp.1.weight"]), p["time_mlp.1.bias"], mv(tm1), act=GELU, out2=mv(tg))
        o.lin_fwd(mv(tg), mv(p["time_mlp.3.weight"]), p["time_mlp.3.bias"], mv(temb))
        o.axpy(1.0, ctx_emb, temb)                                                       # time_emb = time_mlp(noise_level) + context
        o.act_fwd(temb, st, SILU)
        if getattr(self, "cat", None) is None:
            self.prepare_weights()
        cat = self.cat
        ada = self.f(B, ADA)
        o.lin_fwd(mv(st), mv(cat["Wada"]), cat["bada"], mv(ada))
        t.update(noise_level=noise_level, tf=tf, tm1=tm1, tg=tg, temb=temb, st=st, ada=ada)
        # ---- inputs (dmt.py:323-377)
        pos = xn[:, 0:3].contiguous()
        X0n = torch.cat([xn[:, 3:9], cond_n[:, 3:9] if cond_n is not None else torch.zeros_like(xn[:, 3:9])], dim=1).contiguous()
        h = self.f(Nn, 256)
        o.lin_fwd(mv(X0n), mv(p["node_emb.weight"]), p["node_emb.bias"], mv(h))
        X0p = self.z(Pp, 68)
        X0p[:, 0:2] = ex
        xs0 = d2c = None
        if cond_n is not None:
            X0p[:, 2:4] = cond_e
            cpos = cond_n[:, 0:3].contiguous()
            xs0, d2c = self.f(Pp), self.f(Pp)
            self._geom_fwd(TL, cpos, ada, ADA_TOP, "dist_layer.", X0p, 68, 4, xs0, d2c)
            adj = torch.empty(Pp, dtype=torch.int32, device=self.dev)
            E._check(lib.dst_adj_bits(E._ptr(cond_e), C.c_int64(cond_e.shape[1]), E._ptr(d2c), C.c_float(self.edge_th), C.c_float(self.cutoff), C.c_int32(Pp),
                                      E._ptr(adj), s()), "dst_adj_bits")
            t.update(cpos=cpos)
        else:
            adj = torch.full((Pp,), 3, dtype=torch.int32, device=self.dev)
        e = self.f(Pp, 64)
        o.lin_fwd(mv(X0p), mv(p["edge_emb.weight"]), p["edge_emb.bias"], mv(e))
        t.update(X0n=X0n, X0p=X0p, xs0=xs0, d2c=d2c, adj=adj, h0=h, e0=e)
        node_hids, edge_hids = [h], [e]
        blocks = []
        ns = bool(int(os.environ.get("DIFFSPECTRA_NODE_STREAM", "1"))) and getattr(o, "main_stream", None) is not None
        sec = o.node_section if ns else contextlib.nullcontext
        # fused row chains: bf16 mode only (their products are bf16 MFMAs; the fp32 mode keeps the per-operation kernels golden G13 / G17 pin)
        fused_chain = bool(o.bf16) and Pp > 0 and os.environ.get("DIFFSPECTRA_FUSED_CHAIN", "1") != "0"
        for i in range(NB):
            bp = f"e_block_{i}."
            a0 = i * ADA_STRIDE
            bt: Dict[str, object] = dict(pos_in=pos, h_in=h, e_in=e)
            ap = bp + "attn_mpnn."
            # node rows, in front of the attention (dmt.py:148; layers.py:131-140): adaLN modulate, q | k | v as one product (the padding
            # columns come out as exact zeros) - on the node stream, beside the pair rows' geometry and embedding
            with sec():
                hn, st_n1 = self.f(Nn, 256), self.f(Nn, 2)
                o.lnmod_fwd(h, 256, TL.node_off, 1, B, ada, a0 + NODE_OFF + 0, a0 + NODE_OFF + 256, hn, st_n1)
                qkv = self.f(Nn, 768)
                o.lin_fwd(mv(hn), mv(cat["Wqkv"][i]), cat["bqkv"][i], mv(qkv))
            # distances + CondGaussian features, edge embedding (dmt.py:136-139)
            if fused_chain:
                # dmt.py:136-139,145-149 + both lin_edge projections as ONE kernel (csrc/ds_train_chain.hip)
                X1, te = self.f(Pp, 128), self.f(Pp, 512)
                outs = dict(X1=X1, te=te)
                if save:
                    outs.update(xs=self.f(Pp), d2=self.f(Pp), e1=self.f(Pp, 64), st=self.f(Pp, 2), en=self.f(Pp, 64))
                o.pair_front_fwd(TL, pos, ada, a0 + DIST_OFF, a0 + EDGE_OFF + 0, a0 + EDGE_OFF + 64, p[bp + "dist_layer.means.weight"],
                                 p[bp + "dist_layer.stds.weight"], e, self.wb["Wee"][i], p[bp + "edge_emb.bias"], self.wb["Wte"][i], outs)
                xs, d2, e1, st_e1, en = (outs.get(k) for k in ("xs", "d2", "e1", "st", "en"))
            else:
                X1, xs, d2 = self.f(Pp, 128), self.f(Pp), self.f(Pp)
                self._geom_fwd(TL, pos, ada, a0 + DIST_OFF, bp + "dist_layer.", X1, 128, 0, xs, d2)
                X1[:, 64:128] = e
                e1 = self.f(Pp, 64)
                o.lin_fwd(mv(X1), mv(p[bp + "edge_emb.weight"]), p[bp + "edge_emb.bias"], mv(e1))
                en, st_e1 = self.f(Pp, 64), self.f(Pp, 2)
                o.lnmod_fwd(e1, 64, TL.pair_off, 1, B, ada, a0 + EDGE_OFF + 0, a0 + EDGE_OFF + 64, en, st_e1)
                te = self.f(Pp, 512)                                     # tanh(lin_edge0 e) | tanh(lin_edge1 e) as one product; columns 252..255 = tanh(0)
                o.lin_fwd(mv(en), mv(cat["Wte"][i]), None, mv(te), act=TANH)
            te0, te1 = te[:, 0:256], te[:, 256:512]
            # attention (layers.py:131-186)
            if ns:
                o.main_wait()                                        # q | k | v
            attn, alpha = self.f(Nn, 256), self.f(max(D, 1), 16)
            E._check(lib.dst_attn_fwd(C.byref(TL.c), E._ptr(qkv), E._ptr(te0), E._ptr(te1), C.c_int64(512), E._ptr(adj), E._ptr(attn), E._ptr(alpha), s()),
                     "dst_attn_fwd")
            # node stream (dmt.py:156-163): node2edge per node first (the pair rows wait for it), then the gated residual and the FF, the
            # node parts of input_lin and the read-out slice
            with sec():
                u = self.f(Nn, 64)
                o.lin_fwd(mv(attn), mv(p[bp + "node2edge_lin.weight"]), None, mv(u))
                ev_u = o.node_event() if ns else None
                if fused_chain:
                    # dmt.py:113-116,158-163,387 + the node parts of input_lin as ONE kernel (csrc/ds_train_chain.hip): the directed rows wait
                    # for `ac` at the end of this chain
                    h_out, ac, rn = self.f(Nn, 256), self.f(Nn, 512), self.f(Nn, 64)
                    outs = dict(h_out=h_out, ac=ac, rn=rn)
                    if save:
                        outs.update(x1=self.f(Nn, 256), st=self.f(Nn, 2), y1=self.f(Nn, 256), f1=self.f(Nn, 512), s1=self.f(Nn, 512), f2=self.f(Nn, 256))
                    o.node_chain_fwd(TL, h, attn, ada, a0 + NODE_OFF + 512, a0 + NODE_OFF + 768, a0 + NODE_OFF + 1024, a0 + NODE_OFF + 1280,
                                     self.wb["F1"][i], p[bp + "ff_linear1.bias"], self.wb["F2"][i], p[bp + "ff_linear2.bias"], self.wb["Wac"][i],
                                     self.wb["Wn"][i], p[f"node_{i}.bias"], (dp, dseed, 4 * i + 0, 4 * i + 1), outs)
                    x1, st_n2, y1, f1, s1, f2 = (outs.get(k) for k in ("x1", "st", "y1", "f1", "s1", "f2"))
                    ev_ac = o.node_event() if ns else None
                else:
                    x1, y1, st_n2 = self.f(Nn, 256), self.f(Nn, 256), self.f(Nn, 2)
                    o.gate_add_fwd(h, attn, 256, TL.node_off, 1, B, ada, a0 + NODE_OFF + 512, x1)
                    o.lnmod_fwd(x1, 256, TL.node_off, 1, B, ada, a0 + NODE_OFF + 768, a0 + NODE_OFF + 1024, y1, st_n2)
                    f1, s1, f2, h_out = self.f(Nn, 512), self.f(Nn, 512), self.f(Nn, 256), self.f(Nn, 256)
                    # dmt.py:114-116: dropout(act(ff_linear1)) and dropout(ff_linear2) where the GEMMs produce them (f1 = pre-activation, kept)
                    o.lin_fwd(mv(y1), mv(p[bp + "ff_linear1.weight"]), p[bp + "ff_linear1.bias"], mv(f1), act=SILU, out2=mv(s1), drop=(dp, dseed, 4 * i + 0, 512))
                    o.lin_fwd(mv(s1), mv(p[bp + "ff_linear2.weight"]), p[bp + "ff_linear2.bias"], mv(f2), drop=(dp, dseed, 4 * i + 1, 256))
                    o.gate_add_fwd(y1, f2, 256, TL.node_off, 1, B, ada, a0 + NODE_OFF + 1280, h_out)
                    ac = self.f(Nn, 512)                                 # h_row | h_col parts of input_lin as one product
                    o.lin_fwd(mv(h_out), mv(cat["Wac"][i]), None, mv(ac))
                    ev_ac = o.node_event() if ns else None
                    rn = self.f(Nn, 64)                                  # per-block read-out features (dmt.py:387)
                    o.lin_fwd(mv(h_out), mv(p[f"node_{i}.weight"]), p[f"node_{i}.bias"], mv(rn))
            # edge stream (dmt.py:156-157,165-169)
            if ns:
                o.main_wait(ev_u)
            Win = p[bp + "equi_update.input_lin.weight"]                       # [256, 640] = [h_row | h_col | e | dist]
            if fused_chain:
                # dmt.py:156-157,165-169,388 + the edge part of input_lin as ONE kernel (csrc/ds_train_chain.hip); a forward without a tape
                # (the self-conditioning pass) writes only what the rest of the forward reads
                e_out, ed, re_ = self.f(Pp, 64), self.f(Pp, 256), self.f(Pp, 16)
                outs = dict(e_out=e_out, ed=ed, ro=re_)
                if save:
                    outs.update(he=self.f(Pp, 64), xe1=self.f(Pp, 64), st=self.f(Pp, 2), ye1=self.f(Pp, 64), f3=self.f(Pp, 128), s3=self.f(Pp, 128),
                                f4=self.f(Pp, 64), X2=self.f(Pp, 128))
                o.pair_chain_fwd(TL, u, p[bp + "node2edge_lin.bias"], e, X1, 128, ada, a0 + EDGE_OFF + 128, a0 + EDGE_OFF + 192, a0 + EDGE_OFF + 256,
                                 a0 + EDGE_OFF + 320, self.wb["W3"][i], p[bp + "ff_linear3.bias"], self.wb["W4"][i],
                                 p[bp + "ff_linear4.bias"], self.wb["Wed"][i], 128, p[bp + "equi_update.input_lin.bias"], self.wb["Wro"][i],
                                 p[f"edge_{i}.bias"], (dp, dseed, 4 * i + 2, 4 * i + 3), outs)
                he, xe1, st_e2, ye1, f3, s3, f4, X2 = (outs.get(k) for k in ("he", "xe1", "st", "ye1", "f3", "s3", "f4", "X2"))
            else:
                he = self.f(Pp, 64)
                E._check(lib.dst_pair_sum_fwd(C.byref(TL.c), E._ptr(u), C.c_int32(64), E._ptr(p[bp + "node2edge_lin.bias"]), E._ptr(he), s()), "dst_pair_sum_fwd")
                xe1, ye1, st_e2 = self.f(Pp, 64), self.f(Pp, 64), self.f(Pp, 2)
                o.gate_add_fwd(e, he, 64, TL.pair_off, 1, B, ada, a0 + EDGE_OFF + 128, xe1)
                o.lnmod_fwd(xe1, 64, TL.pair_off, 1, B, ada, a0 + EDGE_OFF + 192, a0 + EDGE_OFF + 256, ye1, st_e2)
                f3, s3, f4, e_out = self.f(Pp, 128), self.f(Pp, 128), self.f(Pp, 64), self.f(Pp, 64)
                o.lin_fwd(mv(ye1), mv(p[bp + "ff_linear3.weight"]), p[bp + "ff_linear3.bias"], mv(f3), act=SILU, out2=mv(s3), drop=(dp, dseed, 4 * i + 2, 128))
                o.lin_fwd(mv(s3), mv(p[bp + "ff_linear4.weight"]), p[bp + "ff_linear4.bias"], mv(f4), drop=(dp, dseed, 4 * i + 3, 64))
                o.gate_add_fwd(ye1, f4, 64, TL.pair_off, 1, B, ada, a0 + EDGE_OFF + 320, e_out)
                # equivariant update (dmt.py:37-60) + CoM removal (:385-386)
                X2 = self.f(Pp, 128)
                X2[:, 0:64] = e_out
                X2[:, 64:128] = X1[:, 0:64]
                ed = self.f(Pp, 256)
                o.lin_fwd(mv(X2), mv(Win, 512, 640), p[bp + "equi_update.input_lin.bias"], mv(ed))
                re_ = self.f(Pp, 16)                                     # per-block read-out features (dmt.py:388)
                o.lin_fwd(mv(e_out), mv(p[f"edge_{i}.weight"]), p[f"edge_{i}.bias"], mv(re_))
            if ns:
                o.main_wait(ev_ac)
            if fused_chain:
                # dmt.py:37-48: z of both directions, LayerNorm + modulate, coord_mlp as ONE kernel (csrc/ds_train_chain.hip)
                c2 = self.f(max(D, 1), 3)
                outs = dict(c2=c2)
                if save:
                    outs.update(zz=self.f(D, 256), st=self.f(D, 2), zn=self.f(D, 256), c0=self.f(D, 256), sc0=self.f(D, 256))
                o.dir_chain_fwd(TL, ac, ed, ada, a0 + EQUI_OFF + 0, a0 + EQUI_OFF + 256, self.wb["W0"][i],
                                p[bp + "equi_update.coord_mlp.0.bias"], self.wb["W2"][i], outs)
                zz, st_z, zn, c0, sc0 = (outs.get(k) for k in ("zz", "st", "zn", "c0", "sc0"))
            else:
                zz, zn, st_z = self.f(max(D, 1), 256), self.f(max(D, 1), 256), self.f(max(D, 1), 2)
                E._check(lib.dst_zbuild_fwd(C.byref(TL.c), E._ptr(ac), E._ptr(ed), E._ptr(zz), s()), "dst_zbuild_fwd")
                o.lnmod_fwd(zz, 256, TL.pair_off, 2, B, ada, a0 + EQUI_OFF + 0, a0 + EQUI_OFF + 256, zn, st_z)
                c0, sc0, c2 = self.f(max(D, 1), 256), self.f(max(D, 1), 256), self.f(max(D, 1), 3)
                o.lin_fwd(mv(zn, r1=D), mv(p[bp + "equi_update.coord_mlp.0.weight"]), p[bp + "equi_update.coord_mlp.0.bias"], mv(c0, r1=D), act=SILU,
                          out2=mv(sc0, r1=D))
                o.lin_fwd(mv(sc0, r1=D), mv(p[bp + "equi_update.coord_mlp.2.weight"]), None, mv(c2, r1=D))
            pos_out = self.f(Nn, 3)
            E._check(lib.dst_coord_fwd(C.byref(TL.c), E._ptr(pos), E._ptr(c2), E._ptr(adj), E._ptr(p[bp + "equi_update.coord_norm.scale"]),
                                       E._ptr(pos_out), s()), "dst_coord_fwd")
            node_hids.append(rn)
            edge_hids.append(re_)
            # (kept in every mode: with the node stream nothing a block touched may be freed - and handed out again - before the join below)
            bt.update(X1=X1, xs=xs, d2=d2, e1=e1, hn=hn, st_n1=st_n1, en=en, st_e1=st_e1, qkv=qkv, te=te, attn=attn, alpha=alpha,
                      u=u, he=he, x1=x1, y1=y1, st_n2=st_n2, f1=f1, s1=s1, f2=f2, h_out=h_out, xe1=xe1, ye1=ye1, st_e2=st_e2, f3=f3, s3=s3,
                      f4=f4, e_out=e_out, X2=X2, zz=zz, zn=zn, st_z=st_z, c0=c0, sc0=sc0, c2=c2, ac=ac, ed=ed, rn=rn, re_=re_)
            blocks.append(bt)
            pos, h, e = pos_out, h_out, e_out
        if ns:
            o.main_wait()                                            # join: the last block's node rows and read-out slices
        # ---- read-out MLPs (dmt.py:391-394)
        AH = torch.cat(node_hids, dim=1).contiguous()
        EH = torch.cat(edge_hids, dim=1).contiguous()
        n1, n1s, n2, n2s, atom_pred = self.f(Nn, 256), self.f(Nn, 256), self.f(Nn, 128), self.f(Nn, 128), self.f(Nn, 6)
        o.lin_fwd(mv(AH), mv(p["node_pred_mlp.0.weight"]), p["node_pred_mlp.0.bias"], mv(n1), act=SILU, out2=mv(n1s))
        o.lin_fwd(mv(n1s), mv(p["node_pred_mlp.2.weight"]), p["node_pred_mlp.2.bias"], mv(n2), act=SILU, out2=mv(n2s))
        o.lin_fwd(mv(n2s), mv(p["node_pred_mlp.4.weight"]), p["node_pred_mlp.4.bias"], mv(atom_pred))
        edge_pred = self.f(Pp, 2)
        ro = {}
        for col, name in ((0, "edge_exist_mlp"), (1, "edge_type_mlp")):
            a1, a1s, a2, a2s = self.f(Pp, 64), self.f(Pp, 64), self.f(Pp, 32), self.f(Pp, 32)
            o.lin_fwd(mv(EH), mv(p[name + ".0.weight"]), p[name + ".0.bias"], mv(a1), act=SILU, out2=mv(a1s))
            o.lin_fwd(mv(a1s), mv(p[name + ".2.weight"]), p[name + ".2.bias"], mv(a2), act=SILU, out2=mv(a2s))
            o.lin_fwd(mv(a2s), mv(p[name + ".4.weight"]), p[name + ".4.bias"], mv(edge_pred, col, col + 1))
            ro[name] = (a1, a1s, a2, a2s)
        if save:
            t.update(blocks=blocks, AH=AH, EH=EH, n1=n1, n1s=n1s, n2=n2, n2s=n2s, ro=ro, pos_final=pos)
            self.t = t
        return pos, atom_pred, edge_pred

    # ------------------------------------------------------------------ backward
    def backward(self, dpos, datom, dedge) -> Dict[str, torch.Tensor]:
        """Gradients of every DMT parameter (and ``ctx_emb`` under the key ``'@ctx_emb'``) given the gradients of the three outputs."""
        o, p, lib, t = self.ops, self.p, self.lib, self.t
        TL: TrainLayout = t["TL"]
        B, Nn, Pp = TL.B, TL.Nn, TL.Pp
        D = 2 * Pp
        s = self.ops._s
        ada = t["ada"]
        dp, dseed = t["drop"]
        g: Dict[str, torch.Tensor] = {}

        gbuf = getattr(self, "gbuf", None)
        cat, dcat = self.cat, self.dcat
        self._gw_done = g

        def gw(name):                                       # gradient buffer of a parameter: a view of the trainer's flat stage (zeroed once
            g[name] = gbuf[name] if gbuf is not None else torch.zeros_like(p[name])   # per backward) or, stand-alone, a fresh zero tensor
            return g[name]

        d_ada = self.z(B, ADA)
        o.async_dw = bool(int(os.environ.get("DIFFSPECTRA_ASYNC_DW", "1")))
        # ---- read-out MLPs
        dAH, dEH = self.f(Nn, 768), self.f(Pp, 192)

        def mlp3_bwd(name, x, acts, dy: MV, dx, acc):
            a1, a1s, a2, a2s = acts
            o.lin_bwd_w(dy, mv(a2s), mv(gw(name + ".4.weight")), gw(name + ".4.bias"))
            d2 = torch.empty_like(a2)
            o.lin_bwd_x(dy, mv(p[name + ".4.weight"]), mv(d2), dact=SILU, ref=mv(a2))
            o.lin_bwd_w(mv(d2), mv(a1s), mv(gw(name + ".2.weight")), gw(name + ".2.bias"))
            d1 = torch.empty_like(a1)
            o.lin_bwd_x(mv(d2), mv(p[name + ".2.weight"]), mv(d1), dact=SILU, ref=mv(a1))
            o.lin_bwd_w(mv(d1), mv(x), mv(gw(name + ".0.weight")), gw(name + ".0.bias"))
            o.lin_bwd_x(mv(d1), mv(p[name + ".0.weight"]), mv(dx), acc=acc)

        mlp3_bwd("node_pred_mlp", t["AH"], (t["n1"], t["n1s"], t["n2"], t["n2s"]), mv(datom), dAH, False)
        mlp3_bwd("edge_exist_mlp", t["EH"], t["ro"]["edge_exist_mlp"], mv(dedge, 0, 1), dEH, False)
        mlp3_bwd("edge_type_mlp", t["EH"], t["ro"]["edge_type_mlp"], mv(dedge, 1, 2), dEH, True)
        # ---- blocks, last to first
        dh = self.z(Nn, 256)             # gradient of the block output h (later: block input of the next one)
        de = self.z(Pp, 64)
        dpos_out = dpos
        dd2_buf = self.f(max(Pp, 1))
        ns = bool(int(os.environ.get("DIFFSPECTRA_NODE_STREAM", "1"))) and getattr(o, "main_stream", None) is not None
        sec = o.node_section if ns else contextlib.nullcontext
        fused_chain = bool(o.bf16) and Pp > 0 and os.environ.get("DIFFSPECTRA_FUSED_CHAIN", "1") != "0" and getattr(self, "wb", None) is not None
        # which fused BACKWARD kernels run (DIFFSPECTRA_FUSED_BWD, default all three; the pair- and directed-row kernels take a CU's LDS alone,
        # csrc/ds_train_chain.hip CHAIN_BWD_LDS: sharing a CU with a weight-gradient product they were not bit-reproducible)
        _fb = os.environ.get("DIFFSPECTRA_FUSED_BWD", "node,pair,dir")
        fused_node_b, fused_pair_b, fused_dir_b = (fused_chain and k in _fb for k in ("node", "pair", "dir"))
        for i in reversed(range(NB)):
            bt = t["blocks"][i]
            bp = f"e_block_{i}."
            a0 = i * ADA_STRIDE
            ap = bp + "attn_mpnn."
            # read-out features of this block (node rows on the node stream, here and below: see forward)
            drn, dre = mv(dAH, 256 + 64 * i, 256 + 64 * (i + 1)), mv(dEH, 64 + 16 * i, 64 + 16 * (i + 1))
            with sec():
                o.lin_bwd_w(drn, mv(bt["h_out"]), mv(gw(f"node_{i}.weight")), gw(f"node_{i}.bias"))
                if not fused_node_b:                              # (fused: inside dst_node_chain_bwd)
                    o.lin_bwd_x(drn, mv(p[f"node_{i}.weight"]), mv(dh), acc=True)
            o.lin_bwd_w(dre, mv(bt["e_out"]), mv(gw(f"edge_{i}.weight")), gw(f"edge_{i}.bias"))
            if not fused_pair_b:                                  # (fused: inside dst_pair_chain_bwd)
                o.lin_bwd_x(dre, mv(p[f"edge_{i}.weight"]), mv(de), acc=True)
            # equivariant update
            dpos_in, dc2 = self.f(Nn, 3), self.f(max(D, 1), 3)
            dsp, dms_buf = self.f(B), self.f(B, 128)          # per block: their column sums (parameter gradients) run on the side stream
            E._check(lib.dst_coord_bwd(C.byref(TL.c), E._ptr(bt["pos_in"]), E._ptr(bt["c2"]), E._ptr(t["adj"]), E._ptr(p[bp + "equi_update.coord_norm.scale"]),
                                       E._ptr(dpos_out), E._ptr(dpos_in), E._ptr(dc2), E._ptr(dsp), s()), "dst_coord_bwd")
            o.colsum(mv(dsp.view(B, 1)), gw(bp + "equi_update.coord_norm.scale"), param_grad=True)
            Win = p[bp + "equi_update.input_lin.weight"]
            dWin = gw(bp + "equi_update.input_lin.weight")
            o.lin_bwd_w(mv(dc2, r1=D), mv(bt["sc0"], r1=D), mv(gw(bp + "equi_update.coord_mlp.2.weight")))
            dc0 = self.f(max(D, 1), 256)
            dz = self.f(max(D, 1), 256)                      # (not dc0: the coord_mlp.0 weight gradient may still be reading it on the side stream)
            if fused_dir_b:
                # coord_mlp.2's and coord_mlp.0's input gradients and the LayerNorm backward as ONE kernel (csrc/ds_train_chain.hip)
                o.dir_chain_bwd(TL, dc2, bt["c0"], bt["zz"], bt["st_z"], ada, d_ada, a0 + EQUI_OFF + 0, a0 + EQUI_OFF + 256,
                                p[bp + "equi_update.coord_mlp.2.weight"], self.wb["W0T"][i], dc0, dz)
                o.lin_bwd_w(mv(dc0, r1=D), mv(bt["zn"], r1=D), mv(gw(bp + "equi_update.coord_mlp.0.weight")), gw(bp + "equi_update.coord_mlp.0.bias"))
            else:
                o.lin_bwd_x(mv(dc2, r1=D), mv(p[bp + "equi_update.coord_mlp.2.weight"]), mv(dc0, r1=D), dact=SILU, ref=mv(bt["c0"], r1=D))
                o.lin_bwd_w(mv(dc0, r1=D), mv(bt["zn"], r1=D), mv(gw(bp + "equi_update.coord_mlp.0.weight")), gw(bp + "equi_update.coord_mlp.0.bias"))
                dzn = self.f(max(D, 1), 256)
                o.lin_bwd_x(mv(dc0, r1=D), mv(p[bp + "equi_update.coord_mlp.0.weight"]), mv(dzn, r1=D))
                o.lnmod_bwd(dzn, bt["zz"], bt["st_z"], 256, TL.pair_off, 2, B, ada, d_ada, a0 + EQUI_OFF + 0, a0 + EQUI_OFF + 256, dz, False)
            dac, ded = self.f(Nn, 512), self.f(Pp, 256)
            E._check(lib.dst_zbuild_bwd(C.byref(TL.c), E._ptr(dz), E._ptr(dac), E._ptr(ded), s()), "dst_zbuild_bwd")
            # node stream (the section waits for dac)
            with sec():
                o.lin_bwd_w(mv(dac), mv(bt["h_out"]), mv(dcat["Wac"][i]))           # both node parts at once; scattered into dWin[:, 0:512] at the end
                if fused_node_b:
                    # the five input gradients, both gated residuals and the LayerNorm backward of the node chain as ONE kernel (csrc/ds_train_chain.hip)
                    df2, df1, dh_in, dattn = self.f(Nn, 256), self.f(Nn, 512), self.f(Nn, 256), self.f(Nn, 256)
                    o.node_chain_bwd(TL, dh, dAH.data_ptr() + 4 * (256 + 64 * i), 768, dac, bt["f2"], bt["f1"], bt["x1"], bt["st_n2"], bt["attn"], ada, d_ada,
                                     a0 + NODE_OFF + 512, a0 + NODE_OFF + 768, a0 + NODE_OFF + 1024, a0 + NODE_OFF + 1280, self.wb["WacT"][i], self.wb["WnT"][i],
                                     self.wb["F2T"][i], self.wb["F1T"][i], (dp, dseed, 4 * i + 0, 4 * i + 1), df2, df1, dh_in, dattn)
                    o.lin_bwd_w(mv(df2), mv(bt["s1"]), mv(gw(bp + "ff_linear2.weight")), gw(bp + "ff_linear2.bias"))
                    o.lin_bwd_w(mv(df1), mv(bt["y1"]), mv(gw(bp + "ff_linear1.weight")), gw(bp + "ff_linear1.bias"))
                else:
                    o.lin_bwd_x(mv(dac), mv(cat["Wac"][i]), mv(dh), acc=True)
                    dy1, df2 = self.f(Nn, 256), self.f(Nn, 256)
                    o.gate_add_bwd(dh, bt["f2"], 256, TL.node_off, 1, B, ada, d_ada, a0 + NODE_OFF + 1280, dy1, False, df2, drop=(dp, dseed, 4 * i + 1))
                    o.lin_bwd_w(mv(df2), mv(bt["s1"]), mv(gw(bp + "ff_linear2.weight")), gw(bp + "ff_linear2.bias"))
                    df1 = self.f(Nn, 512)
                    o.lin_bwd_x(mv(df2), mv(p[bp + "ff_linear2.weight"]), mv(df1), dact=SILU, ref=mv(bt["f1"]), drop=(dp, dseed, 4 * i + 0, 512))
                    o.lin_bwd_w(mv(df1), mv(bt["y1"]), mv(gw(bp + "ff_linear1.weight")), gw(bp + "ff_linear1.bias"))
                    o.lin_bwd_x(mv(df1), mv(p[bp + "ff_linear1.weight"]), mv(dy1), acc=True)
                    dx1 = self.f(Nn, 256)
                    o.lnmod_bwd(dy1, bt["x1"], bt["st_n2"], 256, TL.node_off, 1, B, ada, d_ada, a0 + NODE_OFF + 768, a0 + NODE_OFF + 1024, dx1, False)
                    dh_in, dattn = self.f(Nn, 256), self.f(Nn, 256)
                    o.gate_add_bwd(dx1, bt["attn"], 256, TL.node_off, 1, B, ada, d_ada, a0 + NODE_OFF + 512, dh_in, False, dattn)
            # edge stream
            o.lin_bwd_w(mv(ded), mv(bt["X2"]), mv(dWin, 512, 640), gw(bp + "equi_update.input_lin.bias"))
            if fused_pair_b:
                # the five input gradients, both gated residuals and the LayerNorm backward of the rear chain as ONE kernel (csrc/ds_train_chain.hip)
                dfeat2, df4, df3, de_in, dhe = self.f(Pp, 64), self.f(Pp, 64), self.f(Pp, 128), self.f(Pp, 64), self.f(Pp, 64)
                o.pair_chain_bwd(TL, de, dEH.data_ptr() + 4 * (64 + 16 * i), 192, ded, bt["f4"], bt["f3"], bt["xe1"], bt["st_e2"], bt["he"], ada, d_ada,
                                 a0 + EDGE_OFF + 128, a0 + EDGE_OFF + 192, a0 + EDGE_OFF + 256, a0 + EDGE_OFF + 320, self.wb["WedT"][i], self.wb["WroT"][i],
                                 self.wb["W4T"][i], self.wb["W3T"][i], (dp, dseed, 4 * i + 2, 4 * i + 3), dfeat2, df4, df3, de_in, dhe)
                o.lin_bwd_w(mv(df4), mv(bt["s3"]), mv(gw(bp + "ff_linear4.weight")), gw(bp + "ff_linear4.bias"))
                o.lin_bwd_w(mv(df3), mv(bt["ye1"]), mv(gw(bp + "ff_linear3.weight")), gw(bp + "ff_linear3.bias"))
            else:
                o.lin_bwd_x(mv(ded), mv(Win, 512, 576), mv(de), acc=True)
                dfeat2 = self.f(Pp, 64)
                o.lin_bwd_x(mv(ded), mv(Win, 576, 640), mv(dfeat2))
                dye1, df4 = self.f(Pp, 64), self.f(Pp, 64)
                o.gate_add_bwd(de, bt["f4"], 64, TL.pair_off, 1, B, ada, d_ada, a0 + EDGE_OFF + 320, dye1, False, df4, drop=(dp, dseed, 4 * i + 3))
                o.lin_bwd_w(mv(df4), mv(bt["s3"]), mv(gw(bp + "ff_linear4.weight")), gw(bp + "ff_linear4.bias"))
                df3 = self.f(Pp, 128)
                o.lin_bwd_x(mv(df4), mv(p[bp + "ff_linear4.weight"]), mv(df3), dact=SILU, ref=mv(bt["f3"]), drop=(dp, dseed, 4 * i + 2, 128))
                o.lin_bwd_w(mv(df3), mv(bt["ye1"]), mv(gw(bp + "ff_linear3.weight")), gw(bp + "ff_linear3.bias"))
                o.lin_bwd_x(mv(df3), mv(p[bp + "ff_linear3.weight"]), mv(dye1), acc=True)
                dxe1 = self.f(Pp, 64)
                o.lnmod_bwd(dye1, bt["xe1"], bt["st_e2"], 64, TL.pair_off, 1, B, ada, d_ada, a0 + EDGE_OFF + 192, a0 + EDGE_OFF + 256, dxe1, False)
                de_in, dhe = self.f(Pp, 64), self.f(Pp, 64)
                o.gate_add_bwd(dxe1, bt["he"], 64, TL.pair_off, 1, B, ada, d_ada, a0 + EDGE_OFF + 128, de_in, False, dhe)
            # node2edge
            du = self.f(Nn, 64)
            E._check(lib.dst_pair_sum_bwd(C.byref(TL.c), E._ptr(dhe), C.c_int32(64), E._ptr(du), C.c_int32(0), s()), "dst_pair_sum_bwd")
            o.colsum(mv(dhe), gw(bp + "node2edge_lin.bias"), param_grad=True)
            with sec():                                                              # (waits for du)
                o.lin_bwd_w(mv(du), mv(bt["attn"]), mv(gw(bp + "node2edge_lin.weight")))
                o.lin_bwd_x(mv(du), mv(p[bp + "node2edge_lin.weight"]), mv(dattn), acc=True)
            if ns:
                o.main_wait()                                                        # dattn
            # attention
            dqkv, dte = self.f(Nn, 768), self.f(Pp, 512)
            te = bt["te"]
            E._check(lib.dst_attn_bwd(C.byref(TL.c), E._ptr(bt["qkv"]), E._ptr(te[:, 0:256]), E._ptr(te[:, 256:512]), C.c_int64(512), E._ptr(bt["alpha"]),
                                      E._ptr(dattn), E._ptr(dqkv), E._ptr(dte[:, 0:256]), E._ptr(dte[:, 256:512]), C.c_int32(1), E._ptr(o.scratch),
                                      C.c_int64(o.scratch.numel()), s()), "dst_attn_bwd")
            with sec():                                                              # (waits for dqkv) q | k | v and the adaLN modulate of the block input
                dhn = self.f(Nn, 256)
                o.lin_bwd_w(mv(dqkv), mv(bt["hn"]), mv(dcat["Wqkv"][i]), dcat["bqkv"][i])
                o.lin_bwd_x(mv(dqkv), mv(cat["Wqkv"][i]), mv(dhn))
                o.lnmod_bwd(dhn, bt["h_in"], bt["st_n1"], 256, TL.node_off, 1, B, ada, d_ada, a0 + NODE_OFF + 0, a0 + NODE_OFF + 256, dh_in, True)
            o.lin_bwd_w(mv(dte), mv(bt["en"]), mv(dcat["Wte"][i]))                  # lin_edge0 | lin_edge1; dte is already in front of the tanh (te_is_tanh)
            den = self.f(Pp, 64)
            o.lin_bwd_x(mv(dte), mv(cat["Wte"][i]), mv(den))
            de1 = self.f(Pp, 64)
            o.lnmod_bwd(den, bt["e1"], bt["st_e1"], 64, TL.pair_off, 1, B, ada, d_ada, a0 + EDGE_OFF + 0, a0 + EDGE_OFF + 64, de1, False)
            # edge embedding + distance features
            o.lin_bwd_w(mv(de1), mv(bt["X1"]), mv(gw(bp + "edge_emb.weight")), gw(bp + "edge_emb.bias"))
            dfeat1 = self.f(Pp, 64)
            Wee = p[bp + "edge_emb.weight"]
            o.lin_bwd_x(mv(de1), mv(Wee, 0, 64), mv(dfeat1))
            o.lin_bwd_x(mv(de1), mv(Wee, 64, 128), mv(de_in), acc=True)
            self._geom_bwd(TL, bt["pos_in"], ada, d_ada, a0 + DIST_OFF, bp + "dist_layer.", bt["xs"], bt["d2"], dfeat1, dfeat2, dms_buf, dd2_buf, dpos_in)
            o.colsum(mv(dms_buf, 1, 64), gw(bp + "dist_layer.means.weight").view(-1), param_grad=True)      # lane k of the kernel = feature k = Gaussian k - 1
            o.colsum(mv(dms_buf, 65, 128), gw(bp + "dist_layer.stds.weight").view(-1), param_grad=True)
            if ns:
                o.main_wait()           # end of the block: everything the node stream was given precedes what the main stream does next, so this
                                        # block's temporaries may be released (and handed out again) when the next block rebinds their names
            dh, de, dpos_out = dh_in, de_in, dpos_in
        # ---- input embeddings
        o.lin_bwd_w(mv(dh), mv(t["X0n"]), mv(gw("node_emb.weight")), gw("node_emb.bias"))
        o.lin_bwd_w(mv(dAH, 0, 256), mv(t["X0n"]), mv(g["node_emb.weight"]), g["node_emb.bias"], acc=True)
        o.lin_bwd_w(mv(de), mv(t["X0p"]), mv(gw("edge_emb.weight")), gw("edge_emb.bias"))
        o.lin_bwd_w(mv(dEH, 0, 64), mv(t["X0p"]), mv(g["edge_emb.weight"]), g["edge_emb.bias"], acc=True)
        for nm in ("dist_layer.means.weight", "dist_layer.stds.weight"):
            gw(nm)
        if not t["first"]:
            dfeat0 = self.f(Pp, 64)
            o.lin_bwd_x(mv(de), mv(p["edge_emb.weight"], 4, 68), mv(dfeat0))
            o.lin_bwd_x(mv(dEH, 0, 64), mv(p["edge_emb.weight"], 4, 68), mv(dfeat0), acc=True)
            dms_top = self.f(B, 128)                         # (block 0's buffer may still be feeding its column sums on the side stream)
            self._geom_bwd(TL, t["cpos"], ada, d_ada, ADA_TOP, "dist_layer.", t["xs0"], t["d2c"], dfeat0, None, dms_top, dd2_buf, None)
            o.colsum(mv(dms_top, 1, 64), g["dist_layer.means.weight"].view(-1))
            o.colsum(mv(dms_top, 65, 128), g["dist_layer.stds.weight"].view(-1))
        # ---- adaLN table + time embedding
        o.lin_bwd_w(mv(d_ada), mv(t["st"]), mv(dcat["Wada"]), dcat["bada"])
        dtemb = self.f(B, 1024)
        o.lin_bwd_x(mv(d_ada), mv(cat["Wada"]), mv(dtemb), dact=SILU, ref=mv(t["temb"]))
        g["@ctx_emb"] = dtemb
        o.lin_bwd_w(mv(dtemb), mv(t["tg"]), mv(gw("time_mlp.3.weight")), gw("time_mlp.3.bias"))
        dtg = self.f(B, 1024)
        o.lin_bwd_x(mv(dtemb), mv(p["time_mlp.3.weight"]), mv(dtg), dact=GELU, ref=mv(t["tm1"]))
        o.lin_bwd_w(mv(dtg), mv(t["tf"]), mv(gw("time_mlp.1.weight")), gw("time_mlp.1.bias"))
        dtf = self.f(B, 17)
        o.lin_bwd_x(mv(dtg), mv(p["time_mlp.1.weight"]), mv(dtf))
        E._check(lib.dst_time_feat_bwd(E._ptr(t["noise_level"]), E._ptr(p["time_mlp.0.weights"]), E._ptr(dtf), C.c_int32(B),
                                       E._ptr(gw("time_mlp.0.weights")), s()), "dst_time_feat_bwd")
        o.join_dw()                                          # the concatenated gradients are read right here, on the main stream
        o.async_dw = False
        self.scatter_cat_grads(gw)
        self.t = None
        return g

    # ------------------------------------------------------------------ loss
    def loss(self, TL: TrainLayout, pos, atom_pred, edge_pred, tpos, tfeat, tedge, wm, weights=(1.0, 0.25, 0.1)):
        """losses.py:359-394 on packed predictions: (per-molecule loss [B], dpos, datom, dedge)."""
        loss_m, dpos, dfeat, dedge = self.f(TL.B), self.f(TL.Nn, 3), self.f(TL.Nn, 6), self.f(max(TL.Pp, 1), 2)
        E._check(self.lib.dst_loss(C.byref(TL.c), E._ptr(pos), E._ptr(atom_pred), E._ptr(edge_pred), E._ptr(tpos), E._ptr(tfeat), E._ptr(tedge),
                                   E._ptr(wm), C.c_float(weights[0]), C.c_float(weights[1]), C.c_float(weights[2]), E._ptr(loss_m), E._ptr(dpos),
                                   E._ptr(dfeat), E._ptr(dedge), self.ops._s()), "dst_loss")
        return loss_m, dpos, dfeat, dedge
