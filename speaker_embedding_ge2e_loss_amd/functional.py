"""Tensor-level entry points over the C ABI (include/ge2e_hip.h).

torch is used for device memory, the current stream and autograd plumbing only;
all arithmetic of the hot path runs in libge2e_hip.so.  CPU tensors are rejected:
there is no CPU fallback in the product.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib

SMALL_ERR = 1e-6  # hp.general.small_err, strings/constants.py:31
EPS_COS = 1e-8    # F.cosine_similarity default eps (s3:57, s3:70)


def _require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the GE2E HIP path needs a ROCm device tensor "
            "(no CPU fallback exists in speaker_embedding_ge2e_loss_amd)")


def _as_batched(e: torch.Tensor):
    """(N,M,D) -> view (1,N,M,D); (B,N,M,D) unchanged.  Mirrors s3:49-52: must be contiguous."""
    if e.dim() == 3:
        squeeze = True
    elif e.dim() == 4:
        squeeze = False
    else:
        raise ValueError(f"embeddings must be (N,M,D) or (B,N,M,D), got {tuple(e.shape)}")
    if not e.is_contiguous():
        # the reference calls .view() on the input (s3:49,52), which raises for non-contiguous
        raise RuntimeError("embeddings must be contiguous (the reference uses .view(), s3:49-52)")
    if e.dtype != torch.float32:
        raise TypeError(f"embeddings must be float32 at this boundary, got {e.dtype}")
    return (e.unsqueeze(0) if squeeze else e), squeeze


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(t: torch.Tensor) -> int:
    """The raw hipStream_t of torch's current stream on t's device (the C call when this torch has it: 0.3 us against
    5 us for building a torch.cuda.Stream object on every launch)."""
    if _raw_stream is not None:
        idx = t.device.index
        return _raw_stream(idx if idx is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` that costs nothing when dev is already current (the usual case: ~5 us of host time
    per call otherwise, on a path whose kernel takes 30 us)."""
    __slots__ = ("idx", "prev")

    def __init__(self, dev: torch.device):
        self.idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.prev = -1

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


# Per-(shape, device) workspace sizes and per-(device, stream) workspace tensors of the module path.  The size depends on
# the device (its CU count), so the query runs with that device current.  A workspace is reused only by launches on the
# SAME stream, which the stream itself serialises; callers that pass `workspace=` are unaffected.
#  * While the current stream is being CAPTURED the cache is neither read nor written: the workspace comes from the
#    allocator, i.e. from the capturing graph's private pool, which lives as long as the graph.  (A cached pointer baked
#    into a graph would dangle as soon as a later, larger eager call on that stream replaced the cache entry, and two
#    graphs captured on torch's one capture stream would share -- and race on -- one workspace.)
#  * The cache holds at most _WS_CACHE_MAX (device, stream) entries, least recently used out first: a process that keeps
#    creating streams does not keep a workspace (with a large launch's fall-back slices) per dead stream forever.  A
#    dropped or outgrown workspace goes back to the caching allocator, which hands a block out again only in the order
#    of the stream it was allocated on -- the launches still using it are ahead in that very stream.
_ws_bytes_cache: dict = {}
_ws_cache: dict = {}
_WS_CACHE_MAX = 8
_capturing = getattr(torch.cuda, "is_current_stream_capturing", None)


_ws_override: list = []      # innermost `workspace_override` first


class workspace_override:
    """``with workspace_override(ws):`` -- every loss launch inside that does not name a workspace uses ``ws`` (if it is big
    enough and on the right device).  For a caller that OWNS the lifetime question, e.g. a HIP-graph capture: the
    workspace is allocated and initialised once, outside the capture, lives as long as the object that holds the graph,
    and the captured step has no allocation / initialisation node of its own (graphed.GraphedLossStep)."""

    def __init__(self, ws: torch.Tensor):
        self.ws = ws

    def __enter__(self):
        _ws_override.insert(0, self.ws)
        return self.ws

    def __exit__(self, *exc):
        _ws_override.remove(self.ws)
        return False


def _workspace_for(lib, dev: torch.device, stream: int, key: tuple) -> torch.Tensor:
    need = _ws_bytes_cache.get(key)
    if need is None:
        need = _ws_bytes_cache[key] = int(lib.ge2e_workspace_bytes(*key[:6]))
    for ws in _ws_override:
        if ws.device == dev and ws.numel() >= need:
            return ws
    if _capturing is not None and _capturing():
        return alloc_workspace(need, dev)
    k = (key[6], stream)
    ws = _ws_cache.pop(k, None)                 # re-inserted below: dict order = recency
    if ws is None or ws.numel() < need:
        ws = alloc_workspace(need, dev)
        while len(_ws_cache) >= _WS_CACHE_MAX:
            _ws_cache.pop(next(iter(_ws_cache)))
    _ws_cache[k] = ws
    return ws


def workspace_bytes(B: int, N: int, M: int, D: int, variant: str = "softmax", impl: str = "auto") -> int:
    return int(_lib.load().ge2e_workspace_bytes(B, N, M, D, _lib.VARIANTS[variant], _lib.IMPLS[impl]))


def resolve_impl(B: int, N: int, M: int, D: int, variant: str = "softmax", impl: str = "auto") -> str:
    code = _lib.load().ge2e_resolve_impl(B, N, M, D, _lib.VARIANTS[variant], _lib.IMPLS[impl])
    _lib.check(min(code, 0), "ge2e_resolve_impl")
    return _lib.IMPL_NAMES[code]


def alloc_workspace(nbytes: int, device) -> torch.Tensor:
    """A workspace tensor with the team kernel's control block written (ge2e_workspace_init: one small launch on the
    current stream, no sync), so that the first call on it already runs the team kernel.  The block is self-cleaning
    afterwards; an uninitialised workspace would also be safe, its first call would merely take the fall-back."""
    # torch's caching allocator returns >= 512-byte aligned blocks; the library wants 256
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    if ws.is_cuda:
        with torch.cuda.device(ws.device):
            _lib.check(_lib.load().ge2e_workspace_init(ws.data_ptr(), ws.numel(), _stream_ptr(ws)), "ge2e_workspace_init")
    return ws


def _check_scalar_params(w: torch.Tensor, b: torch.Tensor, dev: torch.device):
    """w and b cross the C ABI as raw device pointers to one fp32 each: anything else would be a wild device read."""
    for name, t in (("w", w), ("b", b)):
        _require_cuda(t, name)
        if t.dtype != torch.float32 or t.numel() != 1:
            raise TypeError(f"{name} must be a float32 scalar tensor")
        if t.device != dev:
            raise RuntimeError(f"{name} is on {t.device}, embeddings on {dev}: raw pointers cross the C ABI, all on one device")


def workspace_fallback_count(workspace: torch.Tensor) -> int:
    """Diagnostic (one host sync): how many calls on this workspace were computed by the team kernel's in-call fall-back --
    no team formed, a hand-off timed out beside another stream's kernels, or the control block was not clean -- since
    `alloc_workspace` / `ge2e_workspace_init`.  (TeamCtl.fallbacks, csrc/ge2e_team.hpp: byte 1664 of the workspace.)"""
    return int(workspace[1664:1668].view(torch.int32).item())


@dataclass
class LossOutputs:
    loss: torch.Tensor                  # (B,)
    per: Optional[torch.Tensor]         # (B,N,M)
    dE: Optional[torch.Tensor]          # (B,N,M,D)
    dw: Optional[torch.Tensor]          # (B,)
    db: Optional[torch.Tensor]          # (B,)


def loss_fwd_bwd(embeddings: torch.Tensor, w: torch.Tensor, b: torch.Tensor, *,
                 eps: float = SMALL_ERR, eps_cos: float = EPS_COS, variant: str = "softmax",
                 impl: str = "auto", need_grad: bool = True, need_per: bool = False,
                 out: Optional[LossOutputs] = None,
                 workspace: Optional[torch.Tensor] = None) -> LossOutputs:
    """One enqueue of ge2e_loss_fwd_bwd on the current stream.  No host sync.

    ``out`` / ``workspace`` let a caller (the benchmark, a CUDA-graph capture) reuse
    buffers; otherwise they come from torch's caching allocator.
    """
    lib = _lib.load()
    _require_cuda(embeddings, "embeddings")
    e4, _ = _as_batched(embeddings)
    B, N, M, D = e4.shape
    dev = e4.device
    _check_scalar_params(w, b, dev)
    if out is None:
        f32 = dict(dtype=torch.float32, device=dev)
        sc = torch.empty(3 if need_grad else 1, B, **f32)  # loss | dw | db in one allocation
        out = LossOutputs(
            loss=sc[0],
            per=torch.empty(B, N, M, **f32) if need_per else None,
            dE=torch.empty(B, N, M, D, **f32) if need_grad else None,
            dw=sc[1] if need_grad else None,
            db=sc[2] if need_grad else None)
    v, im = _lib.VARIANTS[variant], _lib.IMPLS[impl]
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    with _on_device(dev) as guard:
        stream = _stream_ptr(e4)
        if workspace is None:
            workspace = _workspace_for(lib, dev, stream, (B, N, M, D, v, im, guard.idx))
        code = lib.ge2e_loss_fwd_bwd(
            e4.data_ptr(), B, N, M, D, w.data_ptr(), b.data_ptr(), eps_cos, eps, v, im,
            out.loss.data_ptr(), ptr(out.per), ptr(out.dE), ptr(out.dw), ptr(out.db),
            workspace.data_ptr(), workspace.numel(), stream)
    _lib.check(code, "ge2e_loss_fwd_bwd")
    return out


# ---- the reference's static helpers (s3:33-38, 41-80, 95-112, 114-127), differentiable like the originals ----------
# Forward AND backward run in libge2e_hip.so; the autograd.Functions below only carry tensors across the C ABI.

def _cos_forward(e4: torch.Tensor, c3: Optional[torch.Tensor], eps: float, eps_cos: float) -> torch.Tensor:
    lib = _lib.load()
    B, N, M, D = e4.shape
    cos = torch.empty(B, N, M, N, dtype=torch.float32, device=e4.device)
    with torch.cuda.device(e4.device):
        if c3 is not None:
            code = lib.ge2e_cos_sim_centroids(e4.data_ptr(), c3.data_ptr(), B, N, M, D, eps_cos, eps, cos.data_ptr(),
                                              _stream_ptr(e4))
            _lib.check(code, "ge2e_cos_sim_centroids")
        else:
            ws = alloc_workspace(lib.ge2e_cos_sim_workspace_bytes(B, N, M, D), e4.device)   # MFMA route where it exists
            code = lib.ge2e_cos_sim(e4.data_ptr(), B, N, M, D, eps_cos, eps, cos.data_ptr(), ws.data_ptr(), ws.numel(),
                                    _stream_ptr(e4))
            _lib.check(code, "ge2e_cos_sim")
    return cos


class _CentroidsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e4):
        lib = _lib.load()
        B, N, M, D = e4.shape
        cent = torch.empty(B, N, D, dtype=torch.float32, device=e4.device)
        with torch.cuda.device(e4.device):
            _lib.check(lib.ge2e_centroids(e4.data_ptr(), B, N, M, D, cent.data_ptr(), _stream_ptr(e4)), "ge2e_centroids")
        ctx.shape = (B, N, M, D)
        return cent

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        B, N, M, D = ctx.shape
        g = g.contiguous().float()
        dE = torch.empty(B, N, M, D, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(lib.ge2e_centroids_bwd(g.data_ptr(), B, N, M, D, dE.data_ptr(), _stream_ptr(g)), "ge2e_centroids_bwd")
        return dE


class _UttCentroidsFunction(torch.autograd.Function):
    """u = (sum - e) / (M - 1): linear and symmetric, so backward is the same kernel on the gradient."""

    @staticmethod
    def _run(x):
        lib = _lib.load()
        B, N, M, D = x.shape
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.ge2e_utterance_centroids(x.data_ptr(), B, N, M, D, out.data_ptr(), _stream_ptr(x)),
                       "ge2e_utterance_centroids")
        return out

    @staticmethod
    def forward(ctx, e4):
        return _UttCentroidsFunction._run(e4)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return _UttCentroidsFunction._run(g.contiguous().float())


class _CosSimFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e4, c3, eps, eps_cos, own=False):
        # own: c3 IS get_centroids(e4) (what every caller of the reference passes) -- the forward then takes ge2e_cos_sim,
        # whose contraction runs on the matrix cores for the shapes the tiled kernel accepts; c3 is kept for the backward
        cos = _cos_forward(e4, None if own else c3, eps, eps_cos)
        ctx.save_for_backward(e4, c3, cos)
        ctx.eps, ctx.eps_cos = eps, eps_cos
        return cos

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        e4, c3, cos = ctx.saved_tensors
        B, N, M, D = e4.shape
        g = g.contiguous().float()
        dE = torch.empty_like(e4)
        dC = torch.empty_like(c3)
        ws = alloc_workspace(lib.ge2e_cos_sim_bwd_workspace_bytes(B, N, M, D), e4.device)
        with torch.cuda.device(e4.device):
            code = lib.ge2e_cos_sim_bwd(e4.data_ptr(), c3.data_ptr(), cos.data_ptr(), g.data_ptr(), B, N, M, D,
                                        ctx.eps_cos, ctx.eps, dE.data_ptr(), dC.data_ptr(), ws.data_ptr(), ws.numel(),
                                        _stream_ptr(e4))
        _lib.check(code, "ge2e_cos_sim_bwd")
        return dE, dC, None, None, None


class _CalcLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s4, eps, variant):
        lib = _lib.load()
        B, N, M, _ = s4.shape
        loss = torch.empty(B, dtype=torch.float32, device=s4.device)
        per = torch.empty(B, N, M, dtype=torch.float32, device=s4.device)
        with torch.cuda.device(s4.device):
            code = lib.ge2e_calc_loss(s4.data_ptr(), B, N, M, eps, _lib.VARIANTS[variant], loss.data_ptr(), per.data_ptr(),
                                      _stream_ptr(s4))
        _lib.check(code, "ge2e_calc_loss")
        ctx.save_for_backward(s4)
        ctx.eps, ctx.variant = eps, variant
        return loss, per

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss, g_per):
        lib = _lib.load()
        (s4,) = ctx.saved_tensors
        B, N, M, _ = s4.shape
        gl = g_loss.contiguous().float() if g_loss is not None else None
        gp = g_per.contiguous().float() if g_per is not None else None
        dS = torch.empty_like(s4)
        with torch.cuda.device(s4.device):
            code = lib.ge2e_calc_loss_bwd(s4.data_ptr(), B, N, M, ctx.eps, _lib.VARIANTS[ctx.variant],
                                          gl.data_ptr() if gl is not None else None,
                                          gp.data_ptr() if gp is not None else None, dS.data_ptr(), _stream_ptr(s4))
        _lib.check(code, "ge2e_calc_loss_bwd")
        return dS, None, None


def cos_sim(embeddings: torch.Tensor, centroids: torch.Tensor | None = None, *, eps: float = SMALL_ERR,
            eps_cos: float = EPS_COS) -> torch.Tensor:
    """get_cos_sim (s3:42-80): (N,M,D) [, centroids (N,D)] -> (N,M,N) or batched; differentiable in both arguments.

    With ``centroids`` the other-speaker columns use them (as the reference does with its second
    argument); without, they are get_centroids(embeddings) -- what every caller in the reference passes.
    """
    _require_cuda(embeddings, "embeddings")
    e4, squeeze = _as_batched(embeddings)
    B, N, M, D = e4.shape
    own = centroids is None
    if own:
        centroids = _CentroidsFunction.apply(e4)
    _require_cuda(centroids, "centroids")
    c3 = centroids.to(torch.float32).reshape(B, -1, D).contiguous()
    if c3.shape[1] != N:
        # s3:77-78 indexes cos_diff[j, :, j] for every speaker j: the reference needs as many centroids as speakers
        raise RuntimeError(f"get_cos_sim: {c3.shape[1]} centroids for {N} speakers")
    cos = _CosSimFunction.apply(e4, c3, float(eps), float(eps_cos), own)
    return cos[0] if squeeze else cos


class _CosSimRowsFunction(torch.autograd.Function):
    """get_cos_sim on the rows of n speakers (columns j0 .. j0 + n - 1) against all N centroids (ge2e_cos_sim_rows)."""

    @staticmethod
    def forward(ctx, e3, c2, j0, eps, eps_cos):
        lib = _lib.load()
        n, M, D = e3.shape
        N = c2.shape[0]
        cos = torch.empty(n, M, N, dtype=torch.float32, device=e3.device)
        with torch.cuda.device(e3.device):
            code = lib.ge2e_cos_sim_rows(e3.data_ptr(), c2.data_ptr(), 1, n, N, j0, M, D, eps_cos, eps, cos.data_ptr(),
                                         _stream_ptr(e3))
        _lib.check(code, "ge2e_cos_sim_rows")
        ctx.save_for_backward(e3, c2, cos)
        ctx.j0, ctx.eps, ctx.eps_cos = j0, eps, eps_cos
        return cos

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        e3, c2, cos = ctx.saved_tensors
        n, M, D = e3.shape
        N = c2.shape[0]
        g = g.contiguous().float()
        dE, dC = torch.empty_like(e3), torch.empty_like(c2)
        ws = torch.empty(max(int(lib.ge2e_cos_sim_rows_bwd_workspace_bytes(1, n, N, M, D)), 256), dtype=torch.uint8, device=e3.device)
        with torch.cuda.device(e3.device):
            code = lib.ge2e_cos_sim_rows_bwd(e3.data_ptr(), c2.data_ptr(), cos.data_ptr(), g.data_ptr(), 1, n, N, ctx.j0, M, D,
                                             ctx.eps_cos, ctx.eps, dE.data_ptr(), dC.data_ptr(), ws.data_ptr(), ws.numel(),
                                             _stream_ptr(e3))
        _lib.check(code, "ge2e_cos_sim_rows_bwd")
        return dE, dC, None, None, None


class _CalcLossRowsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s3, j0, eps, variant):
        lib = _lib.load()
        n, M, N = s3.shape
        loss = torch.empty(1, dtype=torch.float32, device=s3.device)
        per = torch.empty(n, M, dtype=torch.float32, device=s3.device)
        with torch.cuda.device(s3.device):
            code = lib.ge2e_calc_loss_rows(s3.data_ptr(), 1, n, N, j0, M, eps, _lib.VARIANTS[variant], loss.data_ptr(),
                                           per.data_ptr(), _stream_ptr(s3))
        _lib.check(code, "ge2e_calc_loss_rows")
        ctx.save_for_backward(s3)
        ctx.j0, ctx.eps, ctx.variant = j0, eps, variant
        return loss[0], per

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss, g_per):
        lib = _lib.load()
        (s3,) = ctx.saved_tensors
        n, M, N = s3.shape
        gl = g_loss.reshape(1).contiguous().float() if g_loss is not None else None
        gp = g_per.contiguous().float() if g_per is not None else None
        dS = torch.empty_like(s3)
        with torch.cuda.device(s3.device):
            code = lib.ge2e_calc_loss_rows_bwd(s3.data_ptr(), 1, n, N, ctx.j0, M, ctx.eps, _lib.VARIANTS[ctx.variant],
                                               gl.data_ptr() if gl is not None else None,
                                               gp.data_ptr() if gp is not None else None, dS.data_ptr(), _stream_ptr(s3))
        _lib.check(code, "ge2e_calc_loss_rows_bwd")
        return dS, None, None, None


def cos_sim_rows(embeddings: torch.Tensor, centroids: torch.Tensor, first_speaker: int, *, eps: float = SMALL_ERR,
                 eps_cos: float = EPS_COS) -> torch.Tensor:
    """get_cos_sim (s3:42-80) restricted to LOCAL ROWS: ``embeddings`` (n,M,D) are the rows of speakers ``first_speaker`` ..
    ``first_speaker + n - 1`` of a batch whose N centroids are ``centroids`` (N,D) -> (n,M,N); the own-speaker column of
    local speaker jl is ``first_speaker + jl`` and carries the cosine with the leave-one-out centroid of the local rows.
    Differentiable in both arguments (the centroid gradient is this shard's partial one).  SURVEY 8e-ii."""
    _require_cuda(embeddings, "embeddings")
    _require_cuda(centroids, "centroids")
    if embeddings.dim() != 3 or centroids.dim() != 2 or centroids.shape[1] != embeddings.shape[2]:
        raise ValueError(f"cos_sim_rows: embeddings (n,M,D) and centroids (N,D), got {tuple(embeddings.shape)}, {tuple(centroids.shape)}")
    n, N = embeddings.shape[0], centroids.shape[0]
    if first_speaker < 0 or first_speaker + n > N:
        raise ValueError(f"cos_sim_rows: speakers {first_speaker}..{first_speaker + n - 1} of {N}")
    return _CosSimRowsFunction.apply(embeddings.contiguous().float(), centroids.contiguous().float(), int(first_speaker),
                                     float(eps), float(eps_cos))


def calc_loss_rows(sim_rows: torch.Tensor, first_speaker: int, *, eps: float = SMALL_ERR, variant: str = "softmax"):
    """calc_loss (s3:114-127) on the similarity rows (n,M,N) of speakers ``first_speaker`` ..: (sum of the per-row losses,
    per-row losses (n,M)); differentiable."""
    _require_cuda(sim_rows, "sim_rows")
    if sim_rows.dim() != 3:
        raise ValueError(f"sim_rows must be (n,M,N), got {tuple(sim_rows.shape)}")
    return _CalcLossRowsFunction.apply(sim_rows.contiguous().float(), int(first_speaker), float(eps), variant)


def centroids(embeddings: torch.Tensor) -> torch.Tensor:
    """get_centroids (s3:34-38): mean over the utterance axis."""
    _require_cuda(embeddings, "embeddings")
    e4, squeeze = _as_batched(embeddings)
    cent = _CentroidsFunction.apply(e4)
    return cent[0] if squeeze else cent


def utterance_centroids(embeddings: torch.Tensor) -> torch.Tensor:
    """get_utterance_centroids (s3:95-112): leave-one-out centroid of every utterance, (N,M,D) -> (N,M,D)."""
    _require_cuda(embeddings, "embeddings")
    e4, squeeze = _as_batched(embeddings)
    u = _UttCentroidsFunction.apply(e4)
    return u[0] if squeeze else u


def calc_loss(sim_matrix: torch.Tensor, *, eps: float = SMALL_ERR, variant: str = "softmax"):
    """calc_loss (s3:115-127) on a (N,M,N) or (B,N,M,N) similarity matrix: (loss, per_embedding_loss), differentiable."""
    _require_cuda(sim_matrix, "sim_matrix")
    s = sim_matrix
    squeeze = s.dim() == 3
    if squeeze:
        s = s.unsqueeze(0)
    if s.dim() != 4 or s.shape[1] != s.shape[3]:
        raise ValueError(f"sim_matrix must be (N,M,N) or (B,N,M,N), got {tuple(sim_matrix.shape)}")
    loss, per = _CalcLossFunction.apply(s.contiguous().float(), float(eps), variant)
    return (loss[0], per[0]) if squeeze else (loss, per)


class _NormalizeUnpermFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, src, unverified):
        lib = _lib.load()
        rows, D = y.shape
        e = torch.zeros_like(y) if unverified else torch.empty_like(y)
        ctx.unverified = unverified
        rn = torch.empty(rows, dtype=torch.float32, device=y.device)
        with torch.cuda.device(y.device):
            code = lib.ge2e_normalize_unperm(y.data_ptr(), src.data_ptr() if src is not None else None, rows, D,
                                             e.data_ptr(), rn.data_ptr(), _stream_ptr(y))
        _lib.check(code, "ge2e_normalize_unperm")
        ctx.save_for_backward(e, rn, src)
        return e

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        e, rn, src = ctx.saved_tensors
        g = g.contiguous().float()
        rows, D = e.shape
        dy = torch.zeros_like(e) if ctx.unverified else torch.empty_like(e)
        with torch.cuda.device(e.device):
            code = lib.ge2e_normalize_unperm_bwd(g.data_ptr(), e.data_ptr(), rn.data_ptr(),
                                                 src.data_ptr() if src is not None else None, rows, D, dy.data_ptr(),
                                                 _stream_ptr(e))
        _lib.check(code, "ge2e_normalize_unperm_bwd")
        return dy, None, None


def normalize_unperm(y: torch.Tensor, unperm=None, shape=None) -> torch.Tensor:
    """The encoder's tail in one kernel: ``(y / |y|)[unperm]`` -- s2:34 then s4:186 --
    optionally reshaped to ``shape`` = (N, M) -> (N,M,D) (s4:189).  ``y`` (rows, D) is the encoder's
    raw projection; ``unperm`` a permutation of range(rows) (list or int tensor), None = identity.
    Differentiable in ``y``."""
    _require_cuda(y, "y")
    if y.dim() != 2:
        raise ValueError(f"y must be (rows, D), got {tuple(y.shape)}")
    rows = y.shape[0]
    src = _unperm_index(unperm, rows, y.device)
    e = _NormalizeUnpermFunction.apply(y.contiguous().float(), src, _index_is_unverified(unperm))
    if shape is not None:
        e = e.reshape(*shape, e.shape[1])
    return e


def _unperm_index(unperm, rows: int, device) -> Optional[torch.Tensor]:
    """The reference's `unperm` (s4:183-186) as an int32 device tensor.  A list is validated on the host and travels
    through pinned memory (no host sync on the step's critical path).  A TENSOR cannot be validated without a sync, and
    the kernels write every output row exactly once only for a true permutation; so for a tensor the outputs are
    ZERO-initialised (`_index_is_unverified`) and the kernels skip entries outside range(rows): an index that is not a
    permutation yields zero rows / zero gradients where nothing was written, never uninitialised memory and never an
    out-of-range access.  `check_unperm` is the explicit (synchronising) test."""
    if unperm is None:
        return None
    if not torch.is_tensor(unperm):
        if sorted(unperm) != list(range(rows)):
            raise ValueError("unperm must be a permutation of range(rows)")
        return torch.tensor(unperm, dtype=torch.int32).pin_memory().to(device, non_blocking=True)
    if unperm.numel() != rows:
        raise ValueError("unperm must have one entry per row")
    return unperm.to(device=device, dtype=torch.int32).contiguous()


def _index_is_unverified(unperm) -> bool:
    """True for an index the host has not validated (a tensor): outputs indexed through it are zero-initialised."""
    return torch.is_tensor(unperm)


def check_unperm(unperm: torch.Tensor, rows: int) -> None:
    """Raises ValueError unless the tensor is a permutation of range(rows).  One host synchronisation."""
    src = unperm.reshape(-1).long()
    if src.numel() != rows or bool((src < 0).any()) or bool((src >= rows).any()) \
            or bool((torch.bincount(src.clamp(0, rows - 1), minlength=rows) != 1).any()):
        raise ValueError("unperm (tensor) is not a permutation of range(rows)")


def raw_supported(N: int, M: int, D: int) -> bool:
    """Shapes ge2e_loss_raw runs as ONE launch (the one-wave-per-batch kernel's register-only shapes)."""
    return bool(_lib.load().ge2e_raw_supported(N, M, D))


class _GE2ELossRawFunction(torch.autograd.Function):
    """loss(normalize(y)[unperm].view(N,M,D)) and dL/dy in ONE launch (ge2e_loss_fwd_bwd_raw)."""

    @staticmethod
    def forward(ctx, y, src, w, b, N, M, eps, eps_cos, variant, unverified=False):
        lib = _lib.load()
        rows, D = y.shape
        dev = y.device
        need = any(ctx.needs_input_grad[i] for i in (0, 2, 3))
        f32 = dict(dtype=torch.float32, device=dev)
        sc = torch.empty(3, **f32)                                   # loss | dw | db
        dY = (torch.zeros_like(y) if unverified else torch.empty_like(y)) if need else None
        with _on_device(dev):
            code = lib.ge2e_loss_fwd_bwd_raw(
                y.data_ptr(), src.data_ptr() if src is not None else None, 1, N, M, D, w.data_ptr(), b.data_ptr(),
                eps_cos, eps, _lib.VARIANTS[variant], sc.data_ptr(), None, dY.data_ptr() if need else None,
                sc.data_ptr() + 4 if need else None, sc.data_ptr() + 8 if need else None, _stream_ptr(y))
        _lib.check(code, "ge2e_loss_fwd_bwd_raw")
        ctx.w_shape, ctx.b_shape = w.shape, b.shape
        if need:
            ctx.save_for_backward(dY, sc)
        return sc[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        dY, sc = ctx.saved_tensors
        g = grad_out
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        rows, D = dY.shape
        need_y, need_w, need_b = (ctx.needs_input_grad[i] for i in (0, 2, 3))
        gY = torch.empty_like(dY) if need_y else None
        gwb = torch.empty(2, dtype=torch.float32, device=dY.device) if (need_w or need_b) else None
        with _on_device(dY.device):
            code = _lib.load().ge2e_scale_grads(
                dY.data_ptr(), sc.data_ptr() + 4, sc.data_ptr() + 8, g.data_ptr(), 1, 1, rows, 1, D,
                gY.data_ptr() if need_y else None, gwb.data_ptr() if need_w else None,
                gwb.data_ptr() + 4 if need_b else None, _stream_ptr(dY))
        _lib.check(code, "ge2e_scale_grads")
        gw = (gwb[0] if len(ctx.w_shape) == 0 else gwb[0].reshape(ctx.w_shape)) if need_w else None
        gb = (gwb[1] if len(ctx.b_shape) == 0 else gwb[1].reshape(ctx.b_shape)) if need_b else None
        return gY, None, gw, gb, None, None, None, None, None, None


def ge2e_loss_raw(y: torch.Tensor, unperm, w: torch.Tensor, b: torch.Tensor, shape, *, eps: float = SMALL_ERR,
                  eps_cos: float = EPS_COS, variant: str = "softmax") -> torch.Tensor:
    """``GE2ELoss(normalize(y)[unperm].reshape(N, M, D))`` for the encoder's raw projection ``y`` (rows, D) -- s2:34,
    s4:186-189 and s3:19-30 -- as ONE launch that also yields dL/dy.  ``shape`` = (N, M); shapes outside
    ``raw_supported`` take the two-kernel route (normalize_unperm, then ge2e_loss)."""
    _require_cuda(y, "y")
    if y.dim() != 2:
        raise ValueError(f"y must be (rows, D), got {tuple(y.shape)}")
    N, M = int(shape[0]), int(shape[1])
    if N * M != y.shape[0]:
        raise ValueError(f"shape {tuple(shape)} does not match {y.shape[0]} rows")
    if not raw_supported(N, M, y.shape[1]):
        return ge2e_loss(normalize_unperm(y, unperm, shape=(N, M)), w, b, eps=eps, eps_cos=eps_cos, variant=variant)
    src = _unperm_index(unperm, y.shape[0], y.device)
    _check_scalar_params(w, b, y.device)
    y = y.contiguous().float()
    if y.data_ptr() % 16:          # a contiguous view at an odd storage offset: the kernels load 16 bytes per lane
        y = y.clone()
    return _GE2ELossRawFunction.apply(y, src, w, b, N, M, float(eps), float(eps_cos), variant, _index_is_unverified(unperm))


def eer_counts(sim_matrix: torch.Tensor, thresholds) -> torch.Tensor:
    """Integer counts of the calculate_ERR sweep (s5:57-98) for (N,M,N) or (B,N,M,N) similarities:
    -> int32 (T,2) or (B,T,2): [...,0] false accepts (s5:82), [...,1] accepts on the own column (s5:89).
    ``thresholds``: non-decreasing, compared in fp32 as numpy does for a float32 array (s5:58)."""
    _require_cuda(sim_matrix, "sim_matrix")
    s = sim_matrix
    squeeze = s.dim() == 3
    if squeeze:
        s = s.unsqueeze(0)
    if s.dim() != 4 or s.shape[1] != s.shape[3]:
        raise ValueError(f"sim_matrix must be (N,M,N) or (B,N,M,N), got {tuple(sim_matrix.shape)}")
    s = s.contiguous().float()
    thr = torch.as_tensor(thresholds, dtype=torch.float64).to(torch.float32).reshape(-1)
    if thr.numel() < 1 or thr.numel() > 4096 or bool((thr[1:] < thr[:-1]).any()):
        raise ValueError("thresholds must be 1..4096 non-decreasing values")
    thr = thr.to(s.device)
    B, N, M, _ = s.shape
    T = thr.numel()
    counts = torch.empty(B, T, 2, dtype=torch.int32, device=s.device)
    lib = _lib.load()
    with torch.cuda.device(s.device):
        code = lib.ge2e_eer_counts(s.data_ptr(), B, N, M, thr.data_ptr(), T, counts.data_ptr(), _stream_ptr(s))
    _lib.check(code, "ge2e_eer_counts")
    return counts[0] if squeeze else counts


class _GE2ELossFunction(torch.autograd.Function):
    """forward = one fused kernel launch that also produces dE, dw, db;
    backward only scales them by the incoming gradient (no host sync)."""

    @staticmethod
    def forward(ctx, embeddings, w, b, eps, eps_cos, variant, impl):
        need = any(ctx.needs_input_grad[:3])
        squeeze = embeddings.dim() == 3
        # (no .detach(): inside Function.forward nothing is recorded, and only the data pointers cross the boundary)
        o = loss_fwd_bwd(embeddings, w, b, eps=eps, eps_cos=eps_cos, variant=variant, impl=impl, need_grad=need)
        ctx.squeeze = squeeze
        ctx.w_shape, ctx.b_shape = w.shape, b.shape
        if need:
            ctx.save_for_backward(o.dE, o.dw, o.db)
        return o.loss[0] if squeeze else o.loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        dE, dw, db = ctx.saved_tensors
        g = grad_out                                             # 0-dim or (B,)
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        B, N, M, D = dE.shape
        need_e, need_w, need_b = ctx.needs_input_grad[:3]
        # one launch: gE = g dE, gw = sum g dw, gb = sum g db (no host sync; out of place, so a retained graph may run again)
        gE = torch.empty_like(dE) if need_e else None
        gwb = torch.empty(2, dtype=torch.float32, device=dE.device) if (need_w or need_b) else None
        with _on_device(dE.device):
            code = _lib.load().ge2e_scale_grads(
                dE.data_ptr(), dw.data_ptr(), db.data_ptr(), g.data_ptr(), g.numel(), B, N, M, D,
                gE.data_ptr() if need_e else None, gwb.data_ptr() if need_w else None,
                gwb.data_ptr() + 4 if need_b else None, _stream_ptr(dE))
        _lib.check(code, "ge2e_scale_grads")
        if need_e and ctx.squeeze:
            gE = gE[0]
        gw = (gwb[0] if len(ctx.w_shape) == 0 else gwb[0].reshape(ctx.w_shape)) if need_w else None
        gb = (gwb[1] if len(ctx.b_shape) == 0 else gwb[1].reshape(ctx.b_shape)) if need_b else None
        return gE, gw, gb, None, None, None, None


# The autograd node in C++ (libge2e_torch.so, csrc_torch/ge2e_autograd.cpp: torch.ops.ge2e_amd.loss): the same two C-ABI
# calls as _GE2ELossFunction without the Python dispatch around them -- the eager module step at B = 1 is host-bound.
# Used when the library has been built (build.build() does); _GE2ELossFunction is the same node in Python.
_cpp_node = {"tried": False, "op": None, "enabled": True}


def _cpp_loss_op():
    if not _cpp_node["tried"]:
        _cpp_node["tried"] = True
        import os
        import warnings
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libge2e_torch.so")
        # libge2e_torch.so is linked (rpath $ORIGIN) against the IN-TREE libge2e_hip.so: with GE2E_HIP_LIB pointing somewhere
        # else (A/B runs of two libraries) the node would launch the wrong library's kernels -- the Python node is used then
        if os.path.exists(path) and not os.environ.get("GE2E_HIP_LIB"):
            _lib.load()                                   # the core library first: missing -> the loud error, not a dlopen one
            try:
                torch.ops.load_library(path)
                _cpp_node["op"] = torch.ops.ge2e_amd.loss
            except (OSError, RuntimeError) as ex:         # built against another torch / a relinked core library
                warnings.warn(f"libge2e_torch.so could not be loaded ({str(ex)[:120]}); using the Python autograd node "
                              f"(same launches).  Rebuild with `python -m speaker_embedding_ge2e_loss_amd.build --force`.")
    return _cpp_node["op"] if _cpp_node["enabled"] else None


def cpp_node_workspace(like: torch.Tensor):
    """The workspace the C++ autograd node keeps for `like`'s device and the current stream (None when it has none, or when
    the node is not in use): for `workspace_fallback_count`."""
    if _cpp_loss_op() is None:
        return None
    ws = torch.ops.ge2e_amd.cached_workspace(like)
    return ws if ws.numel() > 0 else None


def use_cpp_autograd(enabled: bool) -> None:
    """Choose between the C++ autograd node (default when libge2e_torch.so is built) and the Python one (same launches)."""
    _cpp_node["enabled"] = bool(enabled)


def ge2e_loss(embeddings: torch.Tensor, w: torch.Tensor, b: torch.Tensor, *, eps: float = SMALL_ERR,
              eps_cos: float = EPS_COS, variant: str = "softmax", impl: str = "auto") -> torch.Tensor:
    """Differentiable GE2E loss: 0-dim for (N,M,D) input, (B,) for (B,N,M,D)."""
    _require_cuda(embeddings, "embeddings")
    in_dtype = embeddings.dtype
    if in_dtype != torch.float32:
        # the reference is dtype-generic (s3:19-30 accepts fp16 / fp64 and returns that dtype); the kernels
        # compute in fp32, the casts either side are differentiable torch ops
        embeddings = embeddings.float()
    op = None if _ws_override else _cpp_loss_op()         # (a caller-owned workspace -- a graph capture -- goes through Python)
    if op is not None:
        loss = op(embeddings, w, b, float(eps), float(eps_cos), _lib.VARIANTS[variant], _lib.IMPLS[impl])
    else:
        loss = _GE2ELossFunction.apply(embeddings, w, b, float(eps), float(eps_cos), variant, impl)
    return loss if in_dtype == torch.float32 else loss.to(in_dtype)
