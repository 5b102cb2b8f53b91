"""Tensor-level wrappers over the C ABI (include/occ4d.h): each takes CUDA fp32 /
int32 torch tensors, launches on torch's current stream and returns torch tensors.
torch is plumbing here (device memory + streams); all arithmetic happens in
libocc4d.so.  CPU tensors are rejected -- there is no fallback path."""
import ctypes as C
import math
import os

import warnings

import torch

from . import _lib


class KernelTimer:
    """Times selected launches with HIP events recorded on the launch stream (= torch's current
    stream, which is the stream every libocc4d kernel is launched on) -- bench.py's roofline leg.
    `want(name, **shape)` picks launches; `summary()` synchronises and reports per name."""

    def __init__(self, want):
        self.want = want
        self.events = []

    def launch(self, name, flops, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn()
        e1.record()
        self.events.append((name, e0, e1, float(flops)))
        return rc

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b, f in self.events:
            d = out.setdefault(name, dict(launches=0, total_ms=0.0, total_flops=0.0))
            d['launches'] += 1
            d['total_ms'] += a.elapsed_time(b)
            d['total_flops'] += f
        return out


_timer = None


def set_kernel_timer(t):
    global _timer
    _timer = t


def _launch(name, shape, flops, fn):
    t = _timer
    if t is not None and t.want(name, **shape):
        return t.launch(name, flops, fn)
    return fn()


def stack_batch(tensors):
    """torch.stack over the batch dimension; the usual B == 1 case is a view (torch.stack would copy 54 MB per
    (32256, 416) activation -- four such copies per decode mini-batch)."""
    tensors = list(tensors)
    return tensors[0][None] if len(tensors) == 1 else torch.stack(tensors)


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """The current HIP stream of the current device as a void pointer.  torch.cuda.current_stream() builds a Stream object
    through several Python layers (9 us; a decode mini-batch issues ~40 launches, a training step ~2000): the raw-handle
    accessor behind it is used when this torch build exposes it."""
    if _lib.is_twin():
        return C.c_void_p(0)             # (the CPU twin is synchronous)
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype=torch.float32, name='tensor'):
    if _lib.is_twin():                   # explicit opt-in (cpu_twin.enable()): the g++ twin takes HOST tensors only
        if not isinstance(t, torch.Tensor) or t.is_cuda:
            raise RuntimeError('%s must be a CPU tensor while the CPU twin is loaded' % name)
    elif not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('%s must be a CUDA tensor: occlusions4d_amd runs only on the HIP library '
                           '(no CPU fallback)' % name)
    assert t.dtype == dtype, '%s must be %s, got %s' % (name, dtype, t.dtype)
    return t


def _rows(t, name='tensor'):
    """2-D view whose last dim is contiguous; returns (tensor, row stride in elements)."""
    assert t.dim() == 2, '%s must be 2-D, got %s' % (name, tuple(t.shape))
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _aligned_rows(t, name, k_pad=None):
    """2-D fp32 tensor with 16-byte aligned base and row stride % 4 == 0 (copies if needed).  `k_pad`: the column
    count to zero-pad to (the caller decides it ONCE for both operands of a product, so that an aligned strided view
    with K % 4 != 0 and a freshly padded partner agree on K)."""
    t, ld = _rows(t, name)
    if k_pad is not None and t.shape[1] != k_pad:
        t = torch.nn.functional.pad(t, (0, k_pad - t.shape[1]))        # contiguous copy, ld = k_pad
        ld = k_pad
    if t.data_ptr() % 16 or ld % 4:
        # (a library kernel, not .contiguous(): e.g. the abstract cloud's feature columns, a view 12 bytes into its rows)
        t = copy_rows(t) if ((t.is_cuda or _lib.is_twin()) and t.dtype == torch.float32 and not t.requires_grad) else t.contiguous()
        if t.shape[1] % 4:
            pad = 4 - t.shape[1] % 4
            t = torch.nn.functional.pad(t, (0, pad))
        ld = t.stride(0) if t.shape[0] > 1 else t.shape[1]
        if t.data_ptr() % 16:
            t = t.clone()
    return t, ld


# --------------------------------------------------------------------------------------
# Searches of at least KNN_GRID_MIN_PAIRS query x data pairs (2^26: where the brute-force scan passes ~0.25 ms, twice the grid
# search's latency floor -- profiles/r04_time_knn_grid.txt) over at least KNN_GRID_MIN_DATA points go through the grid
# (OCC4D_KNN_GRID=0: the brute-force kernel everywhere; both give the same lists).
KNN_GRID = os.environ.get('OCC4D_KNN_GRID', '1') != '0'
KNN_GRID_MIN_DATA = int(os.environ.get('OCC4D_KNN_GRID_MIN_DATA', '1024'))
KNN_GRID_MIN_PAIRS = int(os.environ.get('OCC4D_KNN_GRID_MIN_PAIRS', str(1 << 26)))


def knn(query, data, k, metric=0, return_dist=False, int64=False):
    """query (N0,>=3), data (N1,>=3) -> idx (N0,k) [, dist (N0,k)].  metric 0 = squared
    sum (kNN_torch arithmetic), 1 = Euclidean norm (my_knn_torch arithmetic)."""
    q, qs = _rows(_dev(query, name='query'), 'query')
    d, ds = _rows(_dev(data, name='data'), 'data')
    assert q.shape[1] >= 3 and d.shape[1] >= 3
    n0 = q.shape[0]
    idx = torch.empty((n0, k), dtype=torch.int64 if int64 else torch.int32, device=q.device)
    dist = torch.empty((n0, k), dtype=torch.float32, device=q.device) if return_dist else None
    if KNN_GRID and not int64 and d.shape[0] >= KNN_GRID_MIN_DATA and n0 * d.shape[0] >= KNN_GRID_MIN_PAIRS:
        # large searches (the self-kNNs of a cloud, the decoder's query -> abstract-cloud lists of a training step):
        # exact search on a uniform grid of `data`, same lists bit for bit (csrc/gridrad.hip)
        ws = torch.empty(((int(_lib.lib().occ4d_radius_grid_workspace_bytes(d.shape[0])) + 3) // 4,), dtype=torch.float32,
                         device=q.device)
        _lib.check(_lib.lib().occ4d_knn_grid_f32(_ptr(q), qs, n0, _ptr(d), ds, d.shape[0], k, metric, _ptr(idx), _ptr(dist),
                                                 _ptr(ws), _stream()))
        return (idx, dist) if return_dist else idx
    _lib.check(_lib.lib().occ4d_knn_f32(_ptr(q), qs, n0, _ptr(d), ds, d.shape[0], k, metric, _ptr(idx),
                                        1 if int64 else 0, _ptr(dist), _stream()))
    return (idx, dist) if return_dist else idx


def knn_dists(query, data, idx, metric=0):
    """Distances (N0, k) of caller-supplied neighbour lists idx (N0, k) in the search kernel's own expressions: what
    knn(..., return_dist=True) returns beside these indices."""
    q, qs = _rows(_dev(query, name='query'), 'query')
    d, ds = _rows(_dev(data, name='data'), 'data')
    idx = _neighbour_list(idx, q.shape[0], idx.shape[1], 'idx')
    dist = torch.empty(idx.shape, dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().occ4d_knn_dists_f32(_ptr(q), qs, q.shape[0], _ptr(d), ds, d.shape[0], _ptr(idx), idx.shape[1],
                                              metric, _ptr(dist), _stream()))
    return dist


def fps(xyz, m, return_order=False, start=0):
    """xyz (N,>=3), N <= 32768 -> ascending int32 indices (m) of the farthest-point sample that begins at `start`
    [, the selection order (m)]: one workgroup, the cloud in registers."""
    p, ps = _rows(_dev(xyz, name='xyz'), 'xyz')
    out = torch.empty((m,), dtype=torch.int32, device=p.device)
    order = torch.empty((m,), dtype=torch.int32, device=p.device) if return_order else None
    _lib.check(_lib.lib().occ4d_fps_start_f32(_ptr(p), ps, p.shape[0], m, int(start), _ptr(out), _ptr(order),
                                              _stream()))
    return (out, order) if return_order else out


def fps_coop(xyz, m, start=0, n_workgroups=0, return_order=False, check=True):
    """Farthest-point sample over up to 16 cooperating workgroups (n <= 262144), first sample = `start`.
    Returns ascending int32 indices (m) [, selection order (m)].
    A bounded inter-workgroup spin of the kernel can time out (a scheduling delay under a saturating neighbour).  No
    consumer ever sees such a selection: for n <= 32768 (every training cloud) the library enqueues, right behind the
    cooperative kernel, a single-workgroup launch gated on the status word that recomputes the selection (identical
    indices) -- valid in stream order, also inside a captured graph; the status word then reads 2 and the host only
    WARNS.  check=True reads the status word (one 8-byte device->host read): larger clouds (status 1, not repaired)
    are relaunched up to twice before raising.  check=False defers the look to check_pending() without a stall."""
    p, ps = _rows(_dev(xyz, name='xyz'), 'xyz')
    out = torch.empty((m,), dtype=torch.int32, device=p.device)
    order = torch.empty((m,), dtype=torch.int32, device=p.device) if return_order else None
    ws = torch.empty((_lib.lib().occ4d_fps_coop_workspace_bytes() // 8,), dtype=torch.int64, device=p.device)
    for attempt in range(3):
        _lib.check(_lib.lib().occ4d_fps_coop_f32(_ptr(p), ps, p.shape[0], m, int(start), int(n_workgroups), _ptr(out),
                                                 _ptr(order), _ptr(ws), _stream()))
        if not check:
            break
        status = int(ws[-1].item())
        if status == 0:
            break
        if status == 2:
            warnings.warn(_FPS_REPAIRED)
            break
        if attempt == 2:
            raise RuntimeError(_FPS_TIMEOUT)
        warnings.warn('occ4d_fps_coop_f32: an inter-workgroup wait timed out on %d points; relaunching' % p.shape[0])
    if check or torch.cuda.is_current_stream_capturing():
        pass                  # (inside a hipGraph capture: no host-side bookkeeping; the gated repair launch is captured)
    else:
        # deferred check without a stall: status word -> pinned host memory in stream order + an event
        host = torch.empty(1, dtype=torch.int64, pin_memory=True)
        host.copy_(ws[-1:], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _pending_status.append((host, ev))
        if len(_pending_status) > 256:
            check_pending(wait=False)
    return (out, order) if return_order else out


_FPS_TIMEOUT = ('occ4d_fps_coop_f32: an inter-workgroup wait timed out (device oversubscribed?) on a cloud above the '
                'single-workgroup kernel\'s 32768 points; the sampled indices are undefined')
_FPS_REPAIRED = ('occ4d_fps_coop_f32: an inter-workgroup wait timed out (device oversubscribed?); the selection was '
                 'recomputed by the single-workgroup kernel in stream order -- results are valid, the launch was slow')
_pending_status = []


def check_pending(wait=True):
    """Looks at the status word of the fps_coop(..., check=False) launches issued since the last call.  wait=True
    blocks on their completion events; wait=False only looks at launches that have already finished (no stall).
    perform_inference / TrainStep / bench.py call it.  Status 2 (timed out, repaired in stream order: the results every
    consumer saw were valid) -> one warning; status 1 (a cloud above 32768 points launched with check=False, not
    repaired) -> RuntimeError."""
    keep = []
    bad = repaired = False
    for host, ev in _pending_status:
        if wait:
            ev.synchronize()
        elif not ev.query():
            keep.append((host, ev))
            continue
        bad = bad or int(host[0]) == 1
        repaired = repaired or int(host[0]) == 2
    _pending_status[:] = keep
    if repaired:
        warnings.warn(_FPS_REPAIRED)
    if bad:
        raise RuntimeError(_FPS_TIMEOUT)


def fps_coop_debug(spin_limit=0, fail_round=-1):
    """Tests: shorten the kernel's bounded spin (0 = default) / declare a time-out in round `fail_round` (-1 = never)."""
    _lib.check(_lib.lib().occ4d_fps_coop_debug(int(spin_limit), int(fail_round)))


FPS_COOP_MIN_POINTS = 16385     # above the pruned single-workgroup kernel's range the cooperative kernel: since a round
                                # of its all-to-all yields several samples (round 4) it takes 1.09 us per sample at 28672
                                # points with 16 workgroups against the single workgroup's 2.10 (profiles/r04_time_fps_coop.txt)


def fps_auto(xyz, m, start=0, return_order=False):
    """Ascending FPS indices [, the selection order] with the faster kernel for the cloud size: one workgroup with the
    cloud in registers (fps: pruned, several samples per round) below FPS_COOP_MIN_POINTS points, the cooperative
    multi-workgroup kernel above (the training clouds, the dataloader's whole clips).  Beside a training step the
    cooperative kernel is also the cheaper neighbour: sixteen small workgroups cost the step 0.7 ms where one workgroup
    of 8 waves x 224 VGPRs, holding a compute unit, cost 2.2 (profiles/r04_geometry_stream_cost.txt)."""
    n = xyz.shape[0]
    if n < FPS_COOP_MIN_POINTS:
        return fps(xyz, m, start=start, return_order=return_order)
    return fps_coop(xyz, m, start=start, n_workgroups=16, check=False, return_order=return_order)


def linear(x, w, b=None, relu_in=False, relu_out=False, residual=None, out=None,
           add_rows=None, add_div=1, sub_rows=None, sub_idx=None):
    """y = [relu]( [relu](x) @ w.T + b + add_rows[row // add_div] - sub_rows[sub_idx[row]] ) + residual."""
    assert x.dim() == 2 and w.dim() == 2 and w.shape[1] == x.shape[1], \
        'linear: x is %s but w is %s' % (tuple(x.shape), tuple(w.shape))
    k4 = (x.shape[1] + 3) // 4 * 4             # K of the kernel: decided once, applied to both operands
    x, ldx = _aligned_rows(_dev(x, name='x'), 'x', k4)
    w, ldw = _aligned_rows(_dev(w, name='w'), 'w', k4)
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    y, ldy = _rows(_dev(out, name='out'), 'out')
    assert y is out and y.shape == (M, N)
    a = _lib.LinearArgs()
    a.x, a.ldx, a.w, a.ldw = x.data_ptr(), ldx, w.data_ptr(), ldw
    a.bias = _dev(b, name='bias').data_ptr() if b is not None else None
    if b is not None:
        assert b.is_contiguous() and b.numel() == N
    if residual is not None:
        r, ldr = _rows(_dev(residual, name='residual'), 'residual')
        assert r.shape == (M, N)
        a.residual, a.ldr = r.data_ptr(), ldr
    a.y, a.ldy = y.data_ptr(), ldy
    a.M, a.K, a.N = M, K, N
    a.relu_in, a.relu_out = int(relu_in), int(relu_out)
    if add_rows is not None:
        ar, lda = _rows(_dev(add_rows, name='add_rows'), 'add_rows')
        assert ar.shape[1] == N and ar.shape[0] * add_div >= M
        a.add_rows, a.ld_add, a.add_div = ar.data_ptr(), lda, add_div
    if sub_rows is not None:
        sr, lds = _rows(_dev(sub_rows, name='sub_rows'), 'sub_rows')
        si = _dev(sub_idx, torch.int32, 'sub_idx')
        assert sr.shape[1] == N and si.is_contiguous() and si.numel() == M
        a.sub_rows, a.ld_sub, a.sub_idx = sr.data_ptr(), lds, si.data_ptr()
    _lib.check(_launch('linear', dict(M=M, K=K, N=N), 2.0 * M * K * N,
                       lambda: _lib.lib().occ4d_linear_f32(C.byref(a), _stream())))
    return out


def pt_pos_hidden(pos, pos2, idx, P1, c1):
    p, ps = _rows(_dev(pos, name='pos'), 'pos')
    p2, p2s = _rows(_dev(pos2, name='pos2'), 'pos2')
    idx = _dev(idx, torch.int32, 'idx')
    assert idx.is_contiguous() and idx.shape[0] == p.shape[0]
    n, k = idx.shape
    P1 = _dev(P1).contiguous()
    c1 = _dev(c1).contiguous()
    h = P1.shape[0]
    assert P1.shape == (h, 3) and c1.shape == (h,)
    out = torch.empty((n * k, h), dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().occ4d_pt_pos_hidden_f32(_ptr(p), ps, _ptr(p2), p2s, _ptr(idx), n, k, _ptr(P1), _ptr(c1),
                                                  h, _ptr(out), _stream()))
    return out


def pt_attn_in(q, kfeat, pe, idx):
    q, ldq = _rows(_dev(q, name='q'), 'q')
    kf, ldk = _rows(_dev(kfeat, name='kfeat'), 'kfeat')
    n, k = idx.shape
    d = q.shape[1]
    pe = _dev(pe, name='pe')
    assert pe.is_contiguous() and pe.shape == (n * k, d) and kf.shape[1] == d and idx.is_contiguous()
    out = torch.empty((n * k, d), dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().occ4d_pt_attn_in_f32(_ptr(q), ldq, _ptr(kf), ldk, _ptr(pe), _ptr(_dev(idx, torch.int32)),
                                               n, k, d, _ptr(out), _stream()))
    return out


def pt_softmax_agg(logits, v, pe, idx, out=None):
    """agg (n,d) from logits (n*k,d), v (m,d), pe (n*k,d) or None, idx (n,k) int32."""
    n, k = idx.shape
    logits = _dev(logits, name='logits')
    d = logits.shape[1]
    assert logits.is_contiguous() and logits.shape == (n * k, d) and idx.is_contiguous()
    v, ldv = _rows(_dev(v, name='v'), 'v')
    if pe is not None:
        assert pe.is_contiguous() and pe.shape == (n * k, d)
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=logits.device)
    o, ldo = _rows(out, 'out')
    assert o is out
    divisor = float(torch.tensor(math.sqrt(d), dtype=torch.float32))
    _lib.check(_lib.lib().occ4d_pt_softmax_agg_f32(_ptr(logits), _ptr(v), ldv, _ptr(pe),
                                                   _ptr(_dev(idx, torch.int32)), n, k, d, divisor, _ptr(o), ldo,
                                                   _stream()))
    return out


def matmul_f64(a, b):
    """(m, k) @ (k, n) in fp64 on the library (occ4d_matmul_f64); either operand may be a strided (transposed) view."""
    assert a.is_cuda and b.is_cuda and a.dtype == torch.float64 and b.dtype == torch.float64, 'matmul_f64: CUDA fp64 tensors'
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[0]
    m, k = a.shape
    n = b.shape[1]
    c = torch.empty((m, n), dtype=torch.float64, device=a.device)
    _lib.check(_lib.lib().occ4d_matmul_f64(a.data_ptr(), a.stride(0), a.stride(1), b.data_ptr(), b.stride(0), b.stride(1),
                                           c.data_ptr(), m, n, k, _stream()))
    return c


FUSED_ATTN_DIMS = (288, 416)
FUSED_ATTN_MAX_K = 14


X6_ROWLIN_WIDTHS = (208, 416, 832, 1664)


SPLIT_SCHEMES = ('bf16x6', 'f16x3')      # bf16 x 3 pieces, 6 products / fp16 x 2 pieces, 3 products (csrc/bf16x6.hpp)


def pack_rowlin_bf16x6(w, scheme='bf16x6'):
    """(n_out, 416) weight -> the stage-packed piece stream of occ4d_rowlin_bf16x6_f32 (`scheme` 'f16x3':
    occ4d_rowlin_f16x3_f32; the two streams are not interchangeable)."""
    assert scheme in SPLIT_SCHEMES, scheme
    w, ldw = _rows(_dev(w, name='w'), 'w')
    n_out = w.shape[0]
    assert w.shape[1] == 416 and n_out in X6_ROWLIN_WIDTHS
    L = _lib.lib()
    size, pack = ((L.occ4d_rowlin_f16x3_packed_floats, L.occ4d_pack_rowlin_f16x3_f32) if scheme == 'f16x3' else
                  (L.occ4d_rowlin_bf16x6_packed_floats, L.occ4d_pack_rowlin_bf16x6_f32))
    packed = torch.empty((int(size(n_out)),), dtype=torch.float32, device=w.device)
    _lib.check(pack(_ptr(w), ldw, n_out, _ptr(packed), _stream()))
    return packed


def rowlin_bf16x6(x, w, b=None, relu_in=False, res=None, out=None, packed=None, mask=None, res_after_mask=False, n_out=None,
                  scheme='bf16x6'):
    """y = [res +] w [relu](x) + b for a (n_out, 416) weight on the split-precision trunk kernel
    (occ4d_rowlin_bf16x6_f32; n_out in {208, 416, 832, 1664}).  `packed` (pack_rowlin_bf16x6) instead of `w` skips the
    per-call packing.  `mask` (n, n_out): the training data-gradient epilogue (occ4d_rowlin_bf16x6_masked_f32): zero where
    mask <= 0, `res` added before (default) or after the mask.  `scheme` 'f16x3': the fp16 two-piece kernel
    (occ4d_rowlin_f16x3_f32: forward only, no mask)."""
    assert scheme in SPLIT_SCHEMES and not (scheme == 'f16x3' and mask is not None), scheme
    x, ldx = _aligned_rows(_dev(x, name='x'), 'x')
    n = x.shape[0]
    if packed is None:
        packed, n_out = pack_rowlin_bf16x6(w, scheme), w.shape[0]
    assert x.shape[1] == 416 and n_out in X6_ROWLIN_WIDTHS
    L = _lib.lib()
    if out is None:
        out = torch.empty((n, n_out), dtype=torch.float32, device=x.device)
    ldr = 0
    if res is not None:
        res, ldr = _rows(_dev(res, name='res'), 'res')
    bb = _cont(b, 'bias') if b is not None else None
    flops = 2.0 * n * 416 * n_out
    if mask is not None:
        mm, ldm = _rows(_dev(mask, name='mask'), 'mask')
        assert mm.shape == (n, n_out)
        _lib.check(_launch('rowlin', dict(n=n, n_out=n_out), flops, lambda: L.occ4d_rowlin_bf16x6_masked_f32(
            _ptr(x), ldx, _ptr(out), out.stride(0), _ptr(packed), _ptr(bb), n_out, int(relu_in), _ptr(res), ldr,
            int(res_after_mask), _ptr(mm), ldm, n, _stream())))
        return out
    fn = L.occ4d_rowlin_f16x3_f32 if scheme == 'f16x3' else L.occ4d_rowlin_bf16x6_f32
    _lib.check(_launch('rowlin', dict(n=n, n_out=n_out), flops, lambda: fn(
        _ptr(x), ldx, _ptr(out), out.stride(0), _ptr(packed), _ptr(bb), n_out, int(relu_in), _ptr(res), ldr, n, _stream())))
    return out


def implicit_loss_fused(out, target, semantic_classes, density_lw, segmentation_lw, want_grad=True):
    """occ4d_implicit_loss_f32: density BCE-with-logits + masked segmentation cross entropy over (cells, n, g) raw decoder
    outputs (training.implicit_loss's terms with non-zero weight in the published configurations) -> (loss (1,), d loss / d out
    or None), two launches."""
    out, target = _dev(out, name='out'), _dev(target, name='target')
    assert out.dim() == 3 and target.dim() == 3 and out.is_contiguous() and target.is_contiguous()
    cells, n, g = out.shape
    assert tuple(target.shape[:2]) == (cells, n) and target.shape[2] >= 2
    L = _lib.lib()
    ws = torch.empty((int(L.occ4d_implicit_loss_workspace_floats(cells)),), dtype=torch.float32, device=out.device)
    loss = torch.empty((1,), dtype=torch.float32, device=out.device)
    grad = torch.empty_like(out) if want_grad else None
    _lib.check(L.occ4d_implicit_loss_f32(_ptr(out), g, _ptr(target), target.shape[2], cells, n, g, target.shape[2] - 1,
                                         int(semantic_classes), float(density_lw), float(segmentation_lw), _ptr(ws), _ptr(loss),
                                         _ptr(grad), g, _stream()))
    return loss, grad


def resblock_f16x3(x, w0, b0, w1, b1, out=None, packed=None):
    """ResnetBlockFC of width 416 (relu) as one launch in the fp16 two-piece split scheme (occ4d_resblock_f16x3_f32):
    y = x + W1 relu(W0 relu(x) + b0) + b1, the hidden activation in registers.  `out` may be x itself.  `packed`
    (occ4d_pack_resblock_f16x3_f32 of the two weights) instead of w0 / w1 skips the per-call packing."""
    L = _lib.lib()
    x, ldx = _aligned_rows(_dev(x, name='x'), 'x')
    n, d = x.shape
    assert d == TRUNK_WIDTH
    if packed is None:
        w0, w1 = (_cont(t.detach(), 'w') for t in (w0, w1))
        assert tuple(w0.shape) == (d, d) and tuple(w1.shape) == (d, d)
        packed = torch.empty((int(L.occ4d_resblock_f16x3_packed_floats()),), dtype=torch.float32, device=x.device)
        _lib.check(L.occ4d_pack_resblock_f16x3_f32(_ptr(w0), d, _ptr(w1), d, _ptr(packed), _stream()))
    b0, b1 = (_cont(t.detach(), 'b') for t in (b0, b1))
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=x.device)
    o, ldo = _aligned_rows(_dev(out, name='out'), 'out')
    assert o is out and tuple(o.shape) == (n, d)
    flops = 2.0 * 2.0 * n * d * d
    _lib.check(_launch('resblock', dict(n=n), flops, lambda: L.occ4d_resblock_f16x3_f32(
        _ptr(x), ldx, _ptr(o), ldo, _ptr(packed), _ptr(b0), _ptr(b1), n, _stream())))
    return out


def pt_cross_attn(aq, qpos, apos, idx, kt, vt, P1, c1, wp, w2, b2, p2, c2, out=None):
    """Fused vector attention (occ4d_pt_cross_attn_f32): agg (n,d)."""
    aq, ld_aq = _aligned_rows(_dev(aq, name='aq'), 'aq')
    kt, ld_kt = _aligned_rows(_dev(kt, name='kt'), 'kt')
    vt, ld_vt = _rows(_dev(vt, name='vt'), 'vt')
    qp, qs = _rows(_dev(qpos, name='qpos'), 'qpos')
    ap, as_ = _rows(_dev(apos, name='apos'), 'apos')
    idx = _dev(idx, torch.int32, 'idx')
    n, k = idx.shape
    d = vt.shape[1]
    assert idx.is_contiguous() and aq.shape == (n, 2 * d) and kt.shape[1] == 2 * d and qp.shape[0] == n
    ws = [_dev(t).contiguous() for t in (P1, c1, wp, w2, b2, p2, c2)]
    assert ws[2].shape == (2 * d, 32) and ws[3].shape == (d, 2 * d) and ws[5].shape == (d, 32)
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=aq.device)
    o, ldo = _rows(out, 'out')
    assert o is out and o.shape == (n, d)
    divisor = float(torch.tensor(math.sqrt(d), dtype=torch.float32))
    # FLOPs this launch executes (useful, unpadded): per pair Wp (32 x 2d) + W2 (2d x d) + P2 (32 x d)
    flops = 2.0 * n * k * (32 * 2 * d + 2 * d * d + 32 * d)
    fn = _lib.lib().occ4d_pt_cross_attn_f32
    w2_arg, wp_arg = ws[3], ws[2]
    _lib.check(_launch('cross_attn', dict(n=n, k=k, d=d), flops, lambda: fn(
        _ptr(aq), ld_aq, _ptr(qp), qs, _ptr(ap), as_, _ptr(idx), _ptr(kt), ld_kt, _ptr(vt), ld_vt,
        _ptr(ws[0]), _ptr(ws[1]), _ptr(wp_arg), _ptr(w2_arg), _ptr(ws[4]), _ptr(ws[5]), _ptr(ws[6]),
        _ptr(o), ldo, n, kt.shape[0], k, d, divisor, _stream())))
    return out


ATTN16P_SKEW = int(os.environ.get('OCC4D_CA16P_SKEW', '6'))   # phase offset of the paired workgroups (units of s_sleep(127))


def pack_attn16p_stream(w2, b2, wp, p2, c2):
    """Stage-packed weight stream of occ4d_pt_cross_attn16p_f32 (occ4d_pack_attn16p_stream_f32; layout in
    include/occ4d.h): w2 (416, 832) = attn_mlp[2].weight; wp (832, 32) = W1 P2 (merged); p2 (416, 32) = pos_mlp[2].weight
    -> (54, 28 * 256) fp32.  b2 / c2 are not part of the stream: b2 cancels in the softmax over the neighbours, c2 comes
    folded into the value table the kernel is given (vt + c2, see pt_cross_attn16p)."""
    w2, wp, p2 = (_cont(t.detach(), 'w') for t in (w2, wp, p2))
    assert tuple(w2.shape) == (416, 832) and tuple(wp.shape) == (832, 32) and tuple(p2.shape) == (416, 32)
    out = torch.empty((54, 28 * 256), dtype=torch.float32, device=w2.device)
    assert out.numel() == _lib.lib().occ4d_pt_cross_attn16p_stream_floats()
    _lib.check(_lib.lib().occ4d_pack_attn16p_stream_f32(_ptr(w2), _ptr(wp), _ptr(p2), _ptr(out), _stream()))
    return out


def pt_cross_attn16p(aq, qpos, apos, idx, kt, vt, P1, c1, wstream, out=None, skew=None):
    """Fused vector attention, d = 416, paired workgroups (occ4d_pt_cross_attn16p_f32): agg (n, 416).
    `vt` is the value table WITH pos_mlp[2].bias folded in (Wv f + c2), one row per abstract point."""
    aq, ld_aq = _aligned_rows(_dev(aq, name='aq'), 'aq')
    kt, ld_kt = _aligned_rows(_dev(kt, name='kt'), 'kt')
    vt, ld_vt = _rows(_dev(vt, name='vt'), 'vt')
    qp, qs = _rows(_dev(qpos, name='qpos'), 'qpos')
    ap, as_ = _rows(_dev(apos, name='apos'), 'apos')
    idx = _dev(idx, torch.int32, 'idx')
    n, k = idx.shape
    d = vt.shape[1]
    assert idx.is_contiguous() and aq.shape == (n, 2 * d) and kt.shape[1] == 2 * d and qp.shape[0] == n
    ws = [_dev(t).contiguous() for t in (P1, c1)]
    assert ws[0].shape == (32, 3) and wstream.is_contiguous()
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=aq.device)
    o, ldo = _rows(out, 'out')
    assert o is out and o.shape == (n, d)
    divisor = float(torch.tensor(math.sqrt(d), dtype=torch.float32))
    flops = 2.0 * n * k * (32 * 2 * d + 2 * d * d + 32 * d)      # executed, useful (same count as pt_cross_attn)
    sk = ATTN16P_SKEW if skew is None else int(skew)
    _lib.check(_launch('cross_attn', dict(n=n, k=k, d=d), flops, lambda: _lib.lib().occ4d_pt_cross_attn16p_f32(
        _ptr(aq), ld_aq, _ptr(qp), qs, _ptr(ap), as_, _ptr(idx), _ptr(kt), ld_kt, _ptr(vt), ld_vt,
        _ptr(ws[0]), _ptr(ws[1]), _ptr(wstream), _ptr(o), ldo, n, kt.shape[0], k, d, divisor, sk, _stream())))
    return out


def pt_pair_mlp(aq, kt, r, idx, c2, wstream, skew=None, logits=None):
    """Training: the pair tensors of the merged-form layer in one kernel (occ4d_pt_pair_mlp_f32):
    a (n k, 832) = aq_i - kt_j + Wp r before the ReLU, logits (n k, 416) = W2 relu(a) (attn_mlp[2].bias left out: it
    cancels in the softmax), pe (n k, 416) = P2 r + c2.  wstream = pack_attn16p_stream(W2, ., Wp, P2, .).
    `logits`: the (n k, 416) rows the training forward stored (pt_layer_fwd(..., logits_out=)): returned as they are,
    the launch skips GEMM2 and only writes a and pe."""
    aq, ld_aq = _aligned_rows(_dev(aq, name='aq'), 'aq')
    kt, ld_kt = _aligned_rows(_dev(kt, name='kt'), 'kt')
    idx = _dev(idx, torch.int32, 'idx')
    n, k = idx.shape
    d = TRUNK_WIDTH
    r = _dev(r, name='r')
    c2 = _dev(c2).contiguous()
    assert idx.is_contiguous() and r.is_contiguous() and r.shape == (n * k, 32) and aq.shape == (n, 2 * d)
    assert kt.shape[1] == 2 * d and c2.shape == (d,) and wstream.is_contiguous()
    a = torch.empty((n * k, 2 * d), dtype=torch.float32, device=aq.device)
    stored = logits is not None
    if stored:
        logits = _dev(logits, name='logits')
        assert logits.is_contiguous() and tuple(logits.shape) == (n * k, d)
    else:
        logits = torch.empty((n * k, d), dtype=torch.float32, device=aq.device)
    pe = torch.empty((n * k, d), dtype=torch.float32, device=aq.device)
    flops = 2.0 * n * k * (32 * 2 * d + (0 if stored else 2 * d * d) + 32 * d)
    sk = ATTN16P_SKEW if skew is None else int(skew)
    _lib.check(_launch('pair_hidden' if stored else 'pair_mlp', dict(n=n, k=k, d=d), flops,
                       lambda: _lib.lib().occ4d_pt_pair_mlp_f32(
        _ptr(aq), ld_aq, _ptr(kt), ld_kt, _ptr(r), _ptr(idx), _ptr(c2), _ptr(wstream), _ptr(a),
        None if stored else _ptr(logits), _ptr(pe), n, kt.shape[0], k, d, sk, _stream())))
    return a, logits, pe


def pack_attn_bf16x6_stream(w2, wp, p2):
    """Fragment stream of the split-precision attention kernels (occ4d_pack_attn_bf16x6_stream_f32): w2 (416, 832) =
    attn_mlp[2].weight, wp (832, 32) = W1 P2 (merged), p2 (416, 32) = pos_mlp[2].weight."""
    w2, wp, p2 = (_cont(t.detach(), 'w') for t in (w2, wp, p2))
    assert w2.shape == (TRUNK_WIDTH, 2 * TRUNK_WIDTH) and wp.shape == (2 * TRUNK_WIDTH, 32) and p2.shape == (TRUNK_WIDTH, 32)
    L = _lib.lib()
    out = torch.empty((int(L.occ4d_pt_cross_attn_bf16x6_stream_floats()),), dtype=torch.float32, device=w2.device)
    _lib.check(L.occ4d_pack_attn_bf16x6_stream_f32(_ptr(w2), _ptr(wp), _ptr(p2), _ptr(out), _stream()))
    return out


def pt_pair_mlp_bf16x6(aq, kt, r, idx, c2, wstream6):
    """pt_pair_mlp on the three-way split bf16 MFMAs (occ4d_pt_pair_mlp_bf16x6_f32; opt-in, fp32-class).
    wstream6 = pack_attn_bf16x6_stream(W2, Wp, P2)."""
    aq, ld_aq = _aligned_rows(_dev(aq, name='aq'), 'aq')
    kt, ld_kt = _aligned_rows(_dev(kt, name='kt'), 'kt')
    idx = _dev(idx, torch.int32, 'idx')
    n, k = idx.shape
    d = TRUNK_WIDTH
    r = _dev(r, name='r')
    c2 = _dev(c2).contiguous()
    assert idx.is_contiguous() and r.is_contiguous() and r.shape == (n * k, 32) and aq.shape == (n, 2 * d)
    assert kt.shape[1] == 2 * d and c2.shape == (d,) and wstream6.is_contiguous()
    a = torch.empty((n * k, 2 * d), dtype=torch.float32, device=aq.device)
    logits = torch.empty((n * k, d), dtype=torch.float32, device=aq.device)
    pe = torch.empty((n * k, d), dtype=torch.float32, device=aq.device)
    flops = 2.0 * n * k * (32 * 2 * d + 2 * d * d + 32 * d)
    _lib.check(_launch('pair_mlp', dict(n=n, k=k, d=d), flops, lambda: _lib.lib().occ4d_pt_pair_mlp_bf16x6_f32(
        _ptr(aq), ld_aq, _ptr(kt), ld_kt, _ptr(r), _ptr(idx), _ptr(c2), _ptr(wstream6), _ptr(a), _ptr(logits), _ptr(pe),
        n, kt.shape[0], k, d, _stream())))
    return a, logits, pe


def layernorm(x, gamma, beta, eps=1e-5, relu=False, out=None):
    x, ldx = _rows(_dev(x, name='x'), 'x')
    n, d = x.shape
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=x.device)
    o, ldo = _rows(out, 'out')
    assert o is out
    g = _dev(gamma).contiguous() if gamma is not None else None
    b = _dev(beta).contiguous() if beta is not None else None
    _lib.check(_lib.lib().occ4d_layernorm_f32(_ptr(x), ldx, _ptr(g), _ptr(b), eps, int(relu), _ptr(o), ldo, n, d,
                                              _stream()))
    return out


def maxpool_gather(y, idx):
    y, ldy = _rows(_dev(y, name='y'), 'y')
    idx = _dev(idx, torch.int32, 'idx')
    assert idx.is_contiguous() and idx.dim() == 2
    n_out, k = idx.shape
    d = y.shape[1]
    z = torch.empty((n_out, d), dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().occ4d_maxpool_gather_f32(_ptr(y), ldy, _ptr(idx), n_out, k, d, _ptr(z), d, _stream()))
    return z


def copy_rows(src, out=None):
    """Contiguous (or `out`: any row-strided fp32 destination of the same shape) copy of a row-strided 2-D fp32 view, by
    a library kernel (the stride-8 xyz view of a point cloud; the column blocks of the encoder's output)."""
    src, lds = _rows(_dev(src, name='src'), 'src')
    n, d = src.shape
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=src.device)
    o, ldo = _rows(_dev(out, name='out'), 'out')
    assert o is out and tuple(o.shape) == (n, d)
    _lib.check(_lib.lib().occ4d_copy_rows_f32(_ptr(o), ldo, _ptr(src), lds, n, d, _stream()))
    return out


def fill_rows(out, value):
    """out[:, :] = value for a row-strided 2-D fp32 view (the level-id channel of the abstract cloud)."""
    o, ldo = _rows(_dev(out, name='out'), 'out')
    assert o is out
    _lib.check(_lib.lib().occ4d_fill_rows_f32(_ptr(o), ldo, o.shape[0], o.shape[1], float(value), _stream()))
    return out


def nested_fps_level(order, orig, m):
    """(positions (m) int32 ascending in the current cloud, their original indices (m) int32): the first m picks of the
    chain's selection order `order` located in the current cloud whose points have ascending original indices `orig`.
    When picks repeat (fewer distinct points than samples) the distinct positions come first, ascending, and the remaining
    entries repeat the last of them."""
    assert order.dtype == torch.int32 and orig.dtype == torch.int32
    _dev(order, torch.int32, 'order'), _dev(orig, torch.int32, 'orig')
    assert order.is_contiguous() and orig.is_contiguous() and order.numel() >= m
    pos = torch.empty((m,), dtype=torch.int32, device=orig.device)
    nxt = torch.empty((m,), dtype=torch.int32, device=orig.device)
    _lib.check(_lib.lib().occ4d_nested_fps_level_i32(_ptr(order), _ptr(orig), orig.numel(), int(m), _ptr(pos), _ptr(nxt),
                                                     _stream()))
    return pos, nxt


def gather_rows(src, idx, cols=None):
    """out[i, :] = src[idx[i], :cols] (cols None = the whole row)."""
    src, lds = _rows(_dev(src, name='src'), 'src')
    idx = _dev(idx, torch.int32, 'idx').contiguous()
    n_out = idx.numel()
    d = src.shape[1] if cols is None else int(cols)
    assert 0 < d <= src.shape[1]
    out = torch.empty((n_out, d), dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().occ4d_gather_rows_f32(_ptr(src), lds, _ptr(idx), n_out, d, _ptr(out), d, _stream()))
    return out


def mean_rows(x):
    x, ldx = _rows(_dev(x, name='x'), 'x')
    out = torch.empty((x.shape[1],), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().occ4d_mean_rows_f32(_ptr(x), ldx, x.shape[0], x.shape[1], _ptr(out), _stream()))
    return out


def posenc(points, n_freq, base_freq=0.1):
    p, ps = _rows(_dev(points, name='points'), 'points')
    n, c = p.shape
    width = c * (2 * n_freq + 1)
    # row stride padded to a multiple of 4 floats so the Linear that follows can vector-load
    ld = (width + 3) // 4 * 4
    buf = torch.zeros((n, ld), dtype=torch.float32, device=p.device) if ld != width else \
        torch.empty((n, ld), dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().occ4d_posenc_f32(_ptr(p), ps, n, c, n_freq, float(base_freq), _ptr(buf), ld, _stream()))
    return buf[:, :width]


def interp_weights(dist):
    dist = _dev(dist, name='dist')
    assert dist.is_contiguous() and dist.dim() == 2
    w = torch.empty_like(dist)
    _lib.check(_lib.lib().occ4d_interp_weights_f32(_ptr(dist), dist.shape[0], dist.shape[1], _ptr(w), _stream()))
    return w


def interp_add(x, cvec, table, idx, w):
    """x (n,d) += cvec + sum_j w[:,j] * table[idx[:,j]]   (in place)."""
    xx, ldx = _rows(_dev(x, name='x'), 'x')
    assert xx is x
    t, ldt = _rows(_dev(table, name='table'), 'table')
    idx = _dev(idx, torch.int32, 'idx')
    w = _dev(w, name='w')
    n, k = idx.shape
    d = x.shape[1]
    assert idx.is_contiguous() and w.is_contiguous() and w.shape == (n, k) and x.shape[0] == n and t.shape[1] == d
    cv = _dev(cvec).contiguous() if cvec is not None else None
    _lib.check(_lib.lib().occ4d_interp_add_f32(_ptr(x), ldx, _ptr(cv), _ptr(t), ldt, _ptr(idx), _ptr(w), n, k, d,
                                               _stream()))
    return x


TRUNK_WIDTH = 416          # width the row-resident trunk kernels are built for (occ4d_trunk_width)


def _pack(fn_name, w, n_stages_plus_1, stage_floats, n_out=None):
    w, ldw = _rows(_dev(w.detach(), name='w'), 'w')
    out = torch.empty((n_stages_plus_1, stage_floats), dtype=torch.float32, device=w.device)
    fn = getattr(_lib.lib(), fn_name)
    args = (_ptr(w), ldw) + ((n_out,) if n_out is not None else ()) + (_ptr(out), _stream())
    _lib.check(fn(*args))
    return out


def pack_trunk_rows(w):
    """(n_out, 416) weight -> stage-packed stream for occ4d_rowlin_f32 / the first layer of occ4d_resblock_f32
    (occ4d_pack_trunk_rows_f32; include/occ4d.h "rows" packing): per 32-output stage the LDS image of its 52 MFMA
    fragments; one extra stage (a copy of stage 0) at the end."""
    n_out, k = w.shape
    assert k == TRUNK_WIDTH and n_out % 32 == 0, 'pack_trunk_rows: (%d, %d) is not (32 s, %d)' % (n_out, k, TRUNK_WIDTH)
    return _pack('occ4d_pack_trunk_rows_f32', w, n_out // 32 + 1, 13312, n_out)


def pack_trunk_cols(w):
    """(416, 416) second-layer weight of a residual block -> "cols" packing (occ4d_pack_trunk_cols_f32): stage j holds
    the 32 input columns 32 j .. 32 j + 31 of every output row."""
    assert tuple(w.shape) == (TRUNK_WIDTH, TRUNK_WIDTH)
    return _pack('occ4d_pack_trunk_cols_f32', w, 14, 13312)


def pack_trunk4_rows(w):
    """(n_out, 416) weight -> stage-packed stream of the half-CU trunk kernels (occ4d_pack_trunk4_rows_f32): per
    16-output stage the LDS image of its 26 MFMA fragments; one extra stage (a copy of stage 0) at the end."""
    n_out, k = w.shape
    assert k == TRUNK_WIDTH and n_out % 16 == 0, 'pack_trunk4_rows: (%d, %d) is not (16 s, %d)' % (n_out, k, TRUNK_WIDTH)
    return _pack('occ4d_pack_trunk4_rows_f32', w, n_out // 16 + 1, 6656, n_out)


def pack_trunk4_cols(w):
    """(416, 416) second-layer weight of a residual block -> "cols" packing of csrc/trunk4.hip
    (occ4d_pack_trunk4_cols_f32): stage j holds the 16 input columns 16 j .. 16 j + 15 of every output row."""
    assert tuple(w.shape) == (TRUNK_WIDTH, TRUNK_WIDTH)
    return _pack('occ4d_pack_trunk4_cols_f32', w, 27, 6656)


def _interp_args(interp, n):
    """(zconst (416), ztab (M, >= 416) row view, idx (n, k) int32, w (n, k)) -> ctypes argument tuple."""
    if interp is None:
        return (None, None, 0, None, None, 0)
    zconst, ztab, idx, w = interp
    zt, ldz = _rows(_dev(ztab, name='ztab'), 'ztab')
    idx = _dev(idx, torch.int32, 'idx')
    assert zt.shape[1] == TRUNK_WIDTH and idx.is_contiguous() and w.is_contiguous() and idx.shape == w.shape
    assert idx.shape[0] == n
    zc = _dev(zconst).contiguous()
    assert zc.numel() == TRUNK_WIDTH
    return (_ptr(zc), _ptr(zt), ldz, _ptr(idx), _ptr(_dev(w, name='w')), idx.shape[1], (zc, zt))


def resblock(x, w0_packed, b0, w1_packed, b1, out=None, interp=None):
    """out = x + W1 relu(W0 relu(x) + b0) + b1 [+ interpolation term]: one fused kernel (occ4d_resblock_f32)."""
    xx, ldx = _rows(_dev(x, name='x'), 'x')
    n, d = xx.shape
    assert d == TRUNK_WIDTH
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=x.device)
    o, ldo = _rows(_dev(out, name='out'), 'out')
    assert o is out and o.shape == (n, d)
    ia = _interp_args(interp, n)
    b0c, b1c = _dev(b0).contiguous(), _dev(b1).contiguous()
    # the packing tells the kernel generation: 27 stages of 6656 floats = csrc/trunk4.hip, 14 of 13312 = csrc/trunk.hip
    half_cu = w0_packed is not None and w0_packed.numel() == 27 * 6656
    assert w0_packed is None or w1_packed is None or (
        w1_packed.numel() == w0_packed.numel() and (half_cu or w0_packed.numel() == 14 * 13312))
    fn = _lib.lib().occ4d_resblock4_f32 if half_cu else _lib.lib().occ4d_resblock_f32
    _lib.check(_launch('resblock', dict(n=n), 4.0 * n * d * d, lambda: fn(
        _ptr(xx), ldx, _ptr(o), ldo, _ptr(w0_packed), _ptr(b0c), _ptr(w1_packed), _ptr(b1c),
        ia[0], ia[1], ia[2], ia[3], ia[4], ia[5], n, _stream())))
    return out


def rowlin(x, w_packed, b, n_out, relu_in=False, residual=None, out=None, interp=None, mask=None, skip=None):
    """out (n, n_out) = [residual +] W [relu](x) + b [+ interpolation term], K = 416 (occ4d_rowlin_f32); with `mask`
    (n, n_out): rows zeroed where mask <= 0 (occ4d_rowlin_masked_f32: the ReLU mask of a data gradient); with `skip`
    (needs mask, half-CU packing): + skip AFTER the mask (occ4d_rowlin4_masked_skip_f32)."""
    xx, ldx = _rows(_dev(x, name='x'), 'x')
    n, d = xx.shape
    half_cu = n_out % 16 == 0 and w_packed.numel() == (n_out // 16 + 1) * 6656      # csrc/trunk4.hip packing
    assert d == TRUNK_WIDTH and (half_cu or (n_out % 32 == 0 and w_packed.numel() == (n_out // 32 + 1) * 13312))
    if out is None:
        out = torch.empty((n, n_out), dtype=torch.float32, device=x.device)
    o, ldo = _rows(_dev(out, name='out'), 'out')
    assert o is out and o.shape == (n, n_out)
    rr, ldr = (None, 0)
    if residual is not None:
        rr, ldr = _rows(_dev(residual, name='residual'), 'residual')
        assert rr.shape == (n, n_out)
    ia = _interp_args(interp, n)
    assert interp is None or n_out == TRUNK_WIDTH
    bc = _dev(b).contiguous()
    assert bc.numel() == n_out
    if skip is not None:
        assert mask is not None and half_cu and residual is None, 'rowlin: skip needs a mask and the half-CU packing'
        rr, ldr = _rows(_dev(skip, name='skip'), 'skip')
        assert rr.shape == (n, n_out)
    if mask is not None:
        mm, ldm = _rows(_dev(mask, name='mask'), 'mask')
        assert mm.shape == (n, n_out) and interp is None
        mfn = _lib.lib().occ4d_rowlin4_masked_f32 if half_cu else _lib.lib().occ4d_rowlin_masked_f32
        if skip is not None:
            mfn = _lib.lib().occ4d_rowlin4_masked_skip_f32
        _lib.check(_launch('rowlin', dict(n=n, n_out=n_out), 2.0 * n * d * n_out,
                           lambda: mfn(_ptr(xx), ldx, _ptr(o), ldo, _ptr(w_packed), _ptr(bc),
                                                                      n_out, int(relu_in), _ptr(rr), ldr, _ptr(mm), ldm, n,
                                                                      _stream())))
        return out
    fn = _lib.lib().occ4d_rowlin4_f32 if half_cu else _lib.lib().occ4d_rowlin_f32
    _lib.check(_launch('rowlin', dict(n=n, n_out=n_out), 2.0 * n * d * n_out, lambda: fn(
        _ptr(xx), ldx, _ptr(o), ldo, _ptr(w_packed), _ptr(bc), n_out, int(relu_in), _ptr(rr), ldr,
        ia[0], ia[1], ia[2], ia[3], ia[4], ia[5], n, _stream())))
    return out


# --------------------------------------------------------------------------------------
# Path-level entry points (include/occ4d.h, last section; csrc/path.hip): one call = the launch sequence of one
# reference forward.  The nn.Module mirrors build the weight structs from their parameters (reference layout) and call
# these; `prepared` / `scene` buffers are cached by the modules, the per-call workspace comes from torch's caching allocator.
# --------------------------------------------------------------------------------------
PATH_ROW_CHUNK = 32768          # most rows per pass inside the library (csrc/path.hip ROW_CHUNK)


def path_row_chunks(n):
    """The passes the library cuts n query rows into (csrc/path.hip row_step): equal sizes, multiples of 9."""
    passes = -(-n // PATH_ROW_CHUNK)
    balance = os.environ.get('OCC4D_ROW_BALANCE', '1') != '0'
    step = PATH_ROW_CHUNK if passes <= 1 or not balance else (-(-n // passes) + 8) // 9 * 9
    return [min(step, n - lo) for lo in range(0, n, step)]


def _attn_flops(c, k, d):
    return 2.0 * c * k * (32 * 2 * d + 2 * d * d + 32 * d)


def _path_timing(name, flops_per_launch):
    """(occ4d_launch_events struct or None, finish callback) for a path-level call that issues len(flops_per_launch)
    launches of kernel family `name`: the library records the event pairs on its launch stream (include/occ4d.h);
    finish(), called after the library call, books the pairs the call really used into the active timer's `events`
    list ((name, start, end, flops) tuples, the format of KernelTimer / bench_train.StepCounter)."""
    t = _timer
    if t is None or not flops_per_launch or not hasattr(t, 'events'):
        return None, (lambda: None)
    n = len(flops_per_launch)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n)]
    for e in evs:
        e.record()                       # (torch creates the hipEvent_t lazily, at the first record)
    arr = (C.c_void_p * (2 * n))(*[e.cuda_event for e in evs])
    st = _lib.LaunchEvents(events=C.cast(arr, C.POINTER(C.c_void_p)), capacity=n, used=0,
                           kernel=_lib.PROFILE_KINDS[name], reserved=0)
    st._keep = (arr, evs)

    def finish():
        for i in range(st.used):
            t.events.append((name, evs[2 * i], evs[2 * i + 1], float(flops_per_launch[i])))
    return st, finish


def pt_layer_prepare(w, flags, device):
    """prepared buffer of occ4d_pt_layer_prepare_f32 for the weight struct `w` (merged matrices + packed streams)."""
    n = int(_lib.lib().occ4d_pt_layer_prepared_floats(C.byref(w), flags))
    if n < 0:
        _lib.check(_lib.EINVAL)
    prep = torch.empty((n,), dtype=torch.float32, device=device)
    _lib.check(_lib.lib().occ4d_pt_layer_prepare_f32(C.byref(w), _ptr(prep), flags, _stream()))
    return prep


def pt_layer_fwd(w, prepared, x, pos, x2, pos2, k, flags=0, knn_idx=None, out=None, logits_out=None, pair_out=None):
    """occ4d_pt_layer_fwd_f32: PointTransformerLayer / PointTransformerBlock forward of ONE cloud.  x (n, d_in), pos
    (n, >= 3) [, x2 (m, dim2), pos2 (m, >= 3)] -> (n, d_out | dim).  `logits_out` (n k, 416): the training forward
    (occ4d_pt_layer_fwd_logits_f32: the pre-softmax logits of every pair stay in HBM for backward; only the layers
    logits_storable() names).  `pair_out` = (a (n k, 832), pe (n k, 416)) beside logits_out: the other two pair tensors
    too."""
    x, ldx = _aligned_rows(_dev(x, name='x'), 'x')
    p, ps = _rows(_dev(pos, name='pos'), 'pos')
    n = x.shape[0]
    d_out = w.d_out if w.post_w else w.dim
    m, x2p, ldx2, p2, p2s = 0, None, 0, None, 0
    if w.cross:
        x2p, ldx2 = _aligned_rows(_dev(x2, name='x2'), 'x2')
        p2, p2s = _rows(_dev(pos2, name='pos2'), 'pos2')
        m = x2p.shape[0]
        assert p2.shape[0] == m
    assert p.shape[0] == n and p.shape[1] >= 3
    if knn_idx is not None:
        knn_idx = _dev(knn_idx, torch.int32, 'knn_idx')
        assert knn_idx.is_contiguous() and tuple(knn_idx.shape) == (n, k)
    if out is None:
        out = torch.empty((n, d_out), dtype=torch.float32, device=x.device)
    o, ldo = _rows(_dev(out, name='out'), 'out')
    assert o is out and tuple(o.shape) == (n, d_out)
    nws = int(_lib.lib().occ4d_pt_layer_workspace_floats(C.byref(w), n, m, k, flags))
    if nws < 0:
        _lib.check(_lib.EINVAL)
    ws = torch.empty((nws,), dtype=torch.float32, device=x.device)
    fused = w.dim in FUSED_ATTN_DIMS and k <= FUSED_ATTN_MAX_K and w.pos_hidden == 32 and not (flags & _lib.PATH_UNFUSED)
    chunks = path_row_chunks(n) if fused else []
    t = _timer
    ev, finish = (None, lambda: None)
    if t is not None and chunks and t.want('cross_attn', n=chunks[0], k=k, d=w.dim):
        ev, finish = _path_timing('cross_attn', [_attn_flops(c, k, w.dim) for c in chunks])
    if logits_out is not None:
        lg = _dev(logits_out, name='logits_out')
        assert lg.is_contiguous() and tuple(lg.shape) == (n * k, w.dim) and logits_storable(w, k, flags)
        pa = pp = None
        if pair_out is not None:
            pa, pp = (_dev(t, name='pair_out') for t in pair_out)
            assert pa.is_contiguous() and pp.is_contiguous() and tuple(pa.shape) == (n * k, 2 * w.dim) and pp.shape == lg.shape
        _lib.check(_lib.lib().occ4d_pt_layer_fwd_logits_f32(
            C.byref(w), _ptr(prepared), _ptr(x), ldx, _ptr(p), ps, n, _ptr(x2p), ldx2, _ptr(p2), p2s, m, k, _ptr(knn_idx),
            None, _ptr(o), ldo, _ptr(lg), _ptr(pa), _ptr(pp), _ptr(ws), flags, C.byref(ev) if ev is not None else None,
            _stream()))
        finish()
        return out
    _lib.check(_lib.lib().occ4d_pt_layer_fwd_f32(
        C.byref(w), _ptr(prepared), _ptr(x), ldx, _ptr(p), ps, n, _ptr(x2p), ldx2, _ptr(p2), p2s, m, k, _ptr(knn_idx), None,
        _ptr(o), ldo, _ptr(ws), flags, C.byref(ev) if ev is not None else None, _stream()))
    finish()
    return out


def logits_storable(w, k, flags):
    """True for the layers whose forward kernel can leave its logits in HBM (occ4d_pt_layer_fwd_logits_f32): cross
    attention on the fp32 paired-workgroup kernel or the bf16 x 3 split kernel -- dim 416, k <= 14, fused, not fp16."""
    off = _lib.PATH_UNFUSED | _lib.PATH_SPLIT_F16 | _lib.PATH_FIRST_GEN
    return bool(w.cross) and w.dim == TRUNK_WIDTH and k <= FUSED_ATTN_MAX_K and w.pos_hidden == 32 and not (flags & off)


def down_pool_fwd(x, weight, bias, nn_idx, norm=0, gamma=None, beta=None, mean=None, var=None, eps=1e-5):
    """occ4d_down_pool_fwd_f32: z[i] = max_j relu(norm(x W^T + b))[nn_idx[i, j]] (model/modules.py:152-158)."""
    x, ldx = _aligned_rows(_dev(x, name='x'), 'x')
    n, d_in = x.shape
    w = _cont(weight.detach(), 'weight')
    b = _cont(bias.detach(), 'bias')
    d_out = w.shape[0]
    nn_idx = _dev(nn_idx, torch.int32, 'nn_idx')
    assert nn_idx.is_contiguous() and nn_idx.dim() == 2 and w.shape[1] == d_in
    n_new, k = nn_idx.shape
    z = torch.empty((n_new, d_out), dtype=torch.float32, device=x.device)
    ws = torch.empty((n * d_out,), dtype=torch.float32, device=x.device)
    opt = [None if t is None else _cont(t.detach(), 'norm parameter') for t in (gamma, beta, mean, var)]
    _lib.check(_lib.lib().occ4d_down_pool_fwd_f32(_ptr(x), ldx, n, d_in, _ptr(w), _ptr(b), d_out, int(norm), _ptr(opt[0]),
                                                  _ptr(opt[1]), _ptr(opt[2]), _ptr(opt[3]), float(eps), _ptr(nn_idx), n_new,
                                                  k, _ptr(z), d_out, _ptr(ws), _stream()))
    return z


def decoder_prepare(w, flags, device):
    n = int(_lib.lib().occ4d_decoder_prepared_floats(C.byref(w), flags))
    if n < 0:
        _lib.check(_lib.EINVAL)
    prep = torch.empty((n,), dtype=torch.float32, device=device)
    _lib.check(_lib.lib().occ4d_decoder_prepare_f32(C.byref(w), _ptr(prep), flags, _stream()))
    return prep


def decoder_prepare_scene(w, prepared, xyz, feats, fglobal, flags):
    xyz, xs = _rows(_dev(xyz, name='points_abstract'), 'points_abstract')
    feats, ldf = _aligned_rows(_dev(feats, name='features_abstract'), 'features_abstract')
    m = xyz.shape[0]
    assert feats.shape[0] == m
    fg = _cont(fglobal, 'features_global') if fglobal is not None and fglobal.numel() else None
    n = int(_lib.lib().occ4d_decoder_scene_floats(C.byref(w), m))
    if n < 0:
        _lib.check(_lib.EINVAL)
    scene = torch.empty((n,), dtype=torch.float32, device=xyz.device)
    _lib.check(_lib.lib().occ4d_decoder_prepare_scene_f32(C.byref(w), _ptr(prepared), _ptr(xyz), xs, _ptr(feats), ldf,
                                                          _ptr(fg), m, _ptr(scene), flags, _stream()))
    return scene


def _neighbour_list(idx, n, k, name, m=None):
    """Caller-supplied neighbour list -> contiguous (n, k) int32 device tensor (any integer dtype comes in).  `m`: the
    size of the cloud the entries index -- they are forced into [0, m) (the gathers behind them are unchecked; the
    library's inference entry point does the same with its own clamp kernel)."""
    if idx is None:
        return None
    assert torch.is_tensor(idx) and (idx.is_cuda != _lib.is_twin()) and not idx.is_floating_point(), \
        name + ' must be an integer CUDA tensor'
    assert tuple(idx.shape) == (n, k), '%s must have shape (%d, %d), got %s' % (name, n, k, tuple(idx.shape))
    idx = idx.to(torch.int32)
    if m is not None:
        idx = idx.clamp(0, m - 1)
    return idx.contiguous()


def decoder_query_fwd(w, prepared, scene, m, queries, flags=0, out=None, penult=None, want_penult=True,
                      knn_local=None, knn_cross=None):
    """occ4d_decoder_query_fwd_f32 on one mini-batch: queries (n, d_in) -> (out (n, G), penult (n, H) or None).
    knn_local (n, k_local) / knn_cross (n, k_cross): the caller's own neighbour lists of the abstract cloud (the
    reference's tie order on clouds with coincident points), or None = searched by the library."""
    q, qs = _rows(_dev(queries, name='points_query'), 'points_query')
    n = q.shape[0]
    kl = _neighbour_list(knn_local, n, w.k_local, 'knn_local')
    kc = _neighbour_list(knn_cross, n, w.k_cross, 'knn_cross') if w.n_cross > 0 else None
    if out is None:
        out = torch.empty((n, w.d_out), dtype=torch.float32, device=q.device)
    o, ldo = _rows(_dev(out, name='out'), 'out')
    assert o is out and tuple(o.shape) == (n, w.d_out)
    if penult is None and want_penult:
        penult = torch.empty((n, w.d_hidden), dtype=torch.float32, device=q.device)
    ldp = 0
    if penult is not None:
        pp, ldp = _rows(_dev(penult, name='penult'), 'penult')
        assert pp is penult and tuple(pp.shape) == (n, w.d_hidden)
    nws = int(_lib.lib().occ4d_decoder_query_workspace_floats(C.byref(w), n, m, flags))
    if nws < 0:
        _lib.check(_lib.EINVAL)
    ws = torch.empty((nws,), dtype=torch.float32, device=q.device)
    d, k = w.d_hidden, w.k_cross
    fused = (w.n_cross > 0 and d in FUSED_ATTN_DIMS and k <= FUSED_ATTN_MAX_K and w.cross[0].pos_hidden == 32
             and not (flags & _lib.PATH_UNFUSED))
    t = _timer
    ev, finish = (None, lambda: None)
    if t is not None and fused and n and t.want('cross_attn', n=min(n, PATH_ROW_CHUNK), k=k, d=d):
        chunks = path_row_chunks(n)
        ev, finish = _path_timing('cross_attn', [_attn_flops(c, k, d) for c in chunks for _ in range(w.n_cross)])
    elif t is not None and n and t.want('resblock', n=min(n, PATH_ROW_CHUNK)):
        chunks = path_row_chunks(n)
        ev, finish = _path_timing('resblock', [4.0 * c * d * d for c in chunks for _ in range(w.n_blocks)])
    _lib.check(_lib.lib().occ4d_decoder_query_fwd_f32(
        C.byref(w), _ptr(prepared), _ptr(scene), m, _ptr(q), qs, n, _ptr(kl), _ptr(kc), _ptr(o), ldo, _ptr(penult), ldp,
        _ptr(ws), flags, C.byref(ev) if ev is not None else None, _stream()))
    finish()
    return out, penult


def squash(out, ops):
    """In-place per-channel post-op; ops: list of G codes (0 identity, 1 sigmoid, 2 clamp[0,1])."""
    o, ld = _rows(_dev(out, name='out'), 'out')
    assert o is out
    n, g = out.shape
    if n == 0:
        return out
    arr = (C.c_int32 * g)(*[int(v) for v in ops])
    _lib.check(_lib.lib().occ4d_squash_f32(_ptr(out), ld, n, g, arr, _stream()))
    return out


def grid_points(counts, mins, spacings, time_idx, device):
    """(nx*ny*nz, 4) float32 cell-centred query grid generated on the device (x slowest, z fastest)."""
    nx, ny, nz = (int(c) for c in counts)
    out = torch.empty((nx * ny * nz, 4), dtype=torch.float32, device=device)
    _dev(out, name='out')
    _lib.check(_lib.lib().occ4d_grid_points_f32(nx, ny, nz, float(mins[0]), float(spacings[0]), float(mins[1]),
                                                float(spacings[1]), float(mins[2]), float(spacings[2]),
                                                float(time_idx), _ptr(out), _stream()))
    return out


def split_solid_air(points_query, implicit_output, threshold, compress_air=False, n_classes=13):
    """Order-preserving split of the decoded queries by density (channel 0) >= threshold.
    Returns (solid (Ns, 4+G), air (Na, 4+G) or (Na, 5) when compress_air) on the device; one 4-byte
    device->host read (the solid count) sizes the outputs."""
    pts = _cont(points_query, 'points_query')
    o, ld = _rows(_dev(implicit_output, name='implicit_output'), 'implicit_output')
    n, g = o.shape
    assert pts.shape == (n, 4)
    if n == 0:
        return (torch.empty((0, 4 + g), dtype=torch.float32, device=o.device),
                torch.empty((0, 5 if compress_air else 4 + g), dtype=torch.float32, device=o.device))
    nb = (n + 255) // 256
    scratch = torch.empty(nb + 1, dtype=torch.int32, device=o.device)
    st = _stream()
    _lib.check(_lib.lib().occ4d_split_count_f32(_ptr(o), ld, n, float(threshold), _ptr(scratch), _ptr(scratch[nb:]), st))
    n_solid = int(scratch[nb].item())
    solid = torch.empty((n_solid, 4 + g), dtype=torch.float32, device=o.device)
    air = torch.empty((n - n_solid, 5 if compress_air else 4 + g), dtype=torch.float32, device=o.device)
    _lib.check(_lib.lib().occ4d_split_write_f32(_ptr(pts), _ptr(o), ld, n, g, float(threshold), _ptr(scratch),
                                                int(bool(compress_air)), int(n_classes), _ptr(solid), _ptr(air), st))
    return solid, air


class RadiusGrid:
    """Target cloud sorted into grid cells for radius tests (occ4d_radius_grid_build_f32): `far(query, r)` -> float32
    flags, 1.0 where no target lies within r (r <= radius_max).  The decisions are those of knn(k=1, metric=1) followed by
    dist > r."""

    def __init__(self, xyz, radius_max):
        p, ps = _rows(_dev(xyz, name='xyz'), 'xyz')
        assert p.shape[0] >= 1 and p.shape[1] >= 3 and radius_max > 0
        self.radius_max = float(radius_max)
        self.n = p.shape[0]
        nbytes = int(_lib.lib().occ4d_radius_grid_workspace_bytes(self.n))
        self.ws = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=p.device)
        _lib.check(_lib.lib().occ4d_radius_grid_build_f32(_ptr(p), ps, self.n, self.radius_max, _ptr(self.ws), _stream()))

    def far(self, query, radius):
        q, qs = _rows(_dev(query, name='query'), 'query')
        assert q.shape[1] >= 3 and 0 <= radius <= self.radius_max
        out = torch.empty((q.shape[0],), dtype=torch.float32, device=q.device)
        _lib.check(_lib.lib().occ4d_radius_far_f32(_ptr(q), qs, q.shape[0], _ptr(self.ws), float(radius), _ptr(out),
                                                   _stream()))
        return out


def compact_rows(rows, key, threshold, strict=True):
    """Order-preserving selection rows[key > threshold] (>= when not strict) -> (rows kept (n', d), keys kept (n'));
    one 4-byte device->host read (the kept count) sizes the outputs."""
    r, ld = _rows(_dev(rows, name='rows'), 'rows')
    k = _dev(key, name='key')
    assert k.dim() == 1 and k.shape[0] == r.shape[0]
    n, d = r.shape
    if n == 0:
        return r.new_empty((0, d)), k.new_empty((0,))
    lk = k.stride(0) if n > 1 else 1
    nb = (n + 255) // 256
    scratch = torch.empty(nb + 1, dtype=torch.int32, device=r.device)
    st = _stream()
    _lib.check(_lib.lib().occ4d_compact_count_f32(_ptr(k), lk, n, float(threshold), int(strict), _ptr(scratch),
                                                  _ptr(scratch[nb:]), st))
    kept = int(scratch[nb].item())
    out_rows = torch.empty((kept, d), dtype=torch.float32, device=r.device)
    out_key = torch.empty((kept,), dtype=torch.float32, device=r.device)
    _lib.check(_lib.lib().occ4d_compact_rows_f32(_ptr(r), ld, n, d, _ptr(k), lk, float(threshold), int(strict),
                                                 _ptr(scratch), _ptr(out_rows), _ptr(out_key), st))
    return out_rows, out_key


def compact_rows_nosync(rows, key, threshold, strict=True):
    """compact_rows without the device->host read: -> (buffer (n, d) whose first `count` rows are rows[key > threshold] in
    order -- the rest is uninitialised --, count (1,) int32 ON THE DEVICE).  For callers that collect several counts and
    read them in one transfer (the point sampler: one synchronisation per stage instead of one per selection)."""
    r, ld = _rows(_dev(rows, name='rows'), 'rows')
    k = _dev(key, name='key')
    assert k.dim() == 1 and k.shape[0] == r.shape[0]
    n, d = r.shape
    if n == 0:
        return r.new_empty((0, d)), torch.zeros((1,), dtype=torch.int32, device=r.device)
    lk = k.stride(0) if n > 1 else 1
    nb = (n + 255) // 256
    scratch = torch.empty(nb + 1, dtype=torch.int32, device=r.device)
    st = _stream()
    _lib.check(_lib.lib().occ4d_compact_count_f32(_ptr(k), lk, n, float(threshold), int(strict), _ptr(scratch),
                                                  _ptr(scratch[nb:]), st))
    out_rows = torch.empty((n, d), dtype=torch.float32, device=r.device)
    out_key = torch.empty((n,), dtype=torch.float32, device=r.device)
    _lib.check(_lib.lib().occ4d_compact_rows_f32(_ptr(r), ld, n, d, _ptr(k), lk, float(threshold), int(strict),
                                                 _ptr(scratch), _ptr(out_rows), _ptr(out_key), st))
    return out_rows, scratch[nb:nb + 1]


def add_rows(a, b):
    """a + b for two (n, d) tensors (exact fp32 add; the sampler's query = target point + offset)."""
    a, lda = _rows(_dev(a, name='a'), 'a')
    b, ldb = _rows(_dev(b, name='b'), 'b')
    assert a.shape == b.shape
    n, d = a.shape
    out = torch.empty((n, d), dtype=torch.float32, device=a.device)
    if n:
        _lib.check(_lib.lib().occ4d_axpby_f32(_ptr(a), lda, 1.0, _ptr(b), ldb, 1.0, n, d, _ptr(out), d, _stream()))
    return out


# --------------------------------------------------------------------------------------
# backward-pass kernels (include/occ4d.h, "Backward pass")
# --------------------------------------------------------------------------------------
def _cont(t, name='tensor'):
    t = _dev(t, name=name)
    return t if t.is_contiguous() else t.contiguous()


def linear_wgrad(g, x, out=None, accumulate=False, bias=False, relu_x=False):
    """dW (N,K) (+)= g^T x with g (M,N), x (M,K) [x -> relu(x) when relu_x].  bias=True also returns
    db (N) = column sums of g, accumulated from the g tiles the kernel stages anyway: (dW, db)."""
    g = _cont(g, 'g')
    x = _cont(x, 'x')
    M, N = g.shape
    K = x.shape[1]
    assert x.shape[0] == M
    n4, k4 = (N + 3) // 4 * 4, (K + 3) // 4 * 4
    if n4 != N:
        g = torch.nn.functional.pad(g, (0, n4 - N))
    if k4 != K:
        x = torch.nn.functional.pad(x, (0, k4 - K))
    splits, floats = C.c_int(0), C.c_int64(0)
    _lib.check(_lib.lib().occ4d_linear_wgrad_workspace(M, n4, k4, C.byref(splits), C.byref(floats)))
    ws = torch.empty((floats.value,), dtype=torch.float32, device=g.device)
    direct = out is not None and n4 == N and k4 == K and out.is_contiguous()
    dw = out if direct else torch.empty((n4, k4), dtype=torch.float32, device=g.device)
    db = torch.empty((n4,), dtype=torch.float32, device=g.device) if bias else None
    assert not (bias and accumulate), 'linear_wgrad: bias gradient with accumulate is not supported'
    _lib.check(_launch('wgrad', dict(M=M, K=K, N=N), 2.0 * M * K * N,
                       lambda: _lib.lib().occ4d_linear_wgrad_bias_f32(
                           _ptr(g), n4, _ptr(x), k4, M, n4, k4, int(relu_x), _ptr(dw), _ptr(db),
                           int(accumulate and direct), _ptr(ws), splits.value, _stream())))
    if bias:
        db = db[:N]
    if direct:
        return (out, db) if bias else out
    dw = dw[:N, :K]
    if out is not None:
        if accumulate:
            out += dw
        else:
            out.copy_(dw)
        return (out, db) if bias else out
    dw = dw.contiguous()
    return (dw, db) if bias else dw


def colsum(x):
    x, ldx = _rows(_dev(x, name='x'), 'x')
    n, d = x.shape
    chunks = max(1, min(512, n // 256))
    ws = torch.empty((chunks * d,), dtype=torch.float32, device=x.device)
    out = torch.empty((d,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().occ4d_colsum_f32(_ptr(x), ldx, n, d, _ptr(out), 0, _ptr(ws), chunks, _stream()))
    return out


def bn_train_fwd(y, gamma, beta, eps):
    """BatchNorm1d (training mode) + ReLU over the rows of y (n, d): (out, batch mean, biased batch variance)."""
    y, ldy = _rows(_dev(y, name='y'), 'y')
    n, d = y.shape
    mean = torch.empty((d,), dtype=torch.float32, device=y.device)
    var = torch.empty((d,), dtype=torch.float32, device=y.device)
    out = torch.empty((n, d), dtype=torch.float32, device=y.device)
    ws = torch.empty((int(_lib.lib().occ4d_bn_workspace_doubles(n, d)),), dtype=torch.float64, device=y.device)
    _lib.check(_lib.lib().occ4d_bn_train_fwd_f32(_ptr(y), ldy, n, d, _ptr(_cont(gamma.detach(), 'gamma')),
                                                 _ptr(_cont(beta.detach(), 'beta')), float(eps), _ptr(mean), _ptr(var),
                                                 _ptr(out), d, _ptr(ws), _stream()))
    return out, mean, var


def bn_train_bwd(y, g, out, mean, var, gamma, eps):
    """(dx, dgamma, dbeta) of bn_train_fwd."""
    y, ldy = _rows(_dev(y, name='y'), 'y')
    g, ldg = _rows(_dev(g, name='g'), 'g')
    out, ldo = _rows(_dev(out, name='out'), 'out')
    n, d = y.shape
    dx = torch.empty((n, d), dtype=torch.float32, device=y.device)
    dgamma = torch.empty((d,), dtype=torch.float32, device=y.device)
    dbeta = torch.empty((d,), dtype=torch.float32, device=y.device)
    ws = torch.empty((int(_lib.lib().occ4d_bn_workspace_doubles(n, d)),), dtype=torch.float64, device=y.device)
    _lib.check(_lib.lib().occ4d_bn_train_bwd_f32(_ptr(y), ldy, _ptr(g), ldg, _ptr(out), ldo, n, d, _ptr(mean), _ptr(var),
                                                 _ptr(_cont(gamma.detach(), 'gamma')), float(eps), _ptr(dx), d, _ptr(dgamma),
                                                 _ptr(dbeta), _ptr(ws), _stream()))
    return dx, dgamma, dbeta


def swish(x):
    x, ldx = _rows(_dev(x, name='x'), 'x')
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().occ4d_swish_f32(_ptr(x), ldx, x.shape[0], x.shape[1], _ptr(out), x.shape[1], _stream()))
    return out


def swish_bwd(g, x):
    g, ldg = _rows(_dev(g, name='g'), 'g')
    x, ldx = _rows(_dev(x, name='x'), 'x')
    assert g.shape == x.shape
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().occ4d_swish_bwd_f32(_ptr(g), ldg, _ptr(x), ldx, x.shape[0], x.shape[1], _ptr(out), x.shape[1],
                                              _stream()))
    return out


def relu_mask(g, ref):
    g, ldg = _rows(_dev(g, name='g'), 'g')
    ref, ldr = _rows(_dev(ref, name='ref'), 'ref')
    n, d = g.shape
    assert ref.shape == (n, d)
    out = torch.empty((n, d), dtype=torch.float32, device=g.device)
    _lib.check(_lib.lib().occ4d_relu_mask_f32(_ptr(g), ldg, _ptr(ref), ldr, n, d, _ptr(out), d, _stream()))
    return out


# Deterministic mode of the backward pass.  The default scatters / small vector gradients accumulate with fp32 atomics
# (order-dependent rounding, ~1e-7 relative: fine for SGD, but two runs -- or an eager step and its hipGraph replay --
# are not bit-identical).  With DETERMINISTIC on, every such reduction runs in a fixed order: index scatters through a
# stable sort by target row + one thread per (target row, channel) adding its segment in order
# (occ4d_segment_gather_sum_f32), the pos-MLP and LayerNorm parameter gradients through fixed-order two-stage sums.
DETERMINISTIC = os.environ.get('OCC4D_DETERMINISTIC', '0') != '0'


class deterministic:
    """with ops.deterministic(): ... -> fixed-order reductions in every backward kernel."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global DETERMINISTIC
        self._old, DETERMINISTIC = DETERMINISTIC, self.on

    def __exit__(self, *exc):
        global DETERMINISTIC
        DETERMINISTIC = self._old


_SEGMENTS = []       # a few recent (key, base tensor, order, offsets): one neighbour list serves several scatters


def forget_segments():
    """Drops the cached segmentations (see _segments: needed only by callers that refill an index buffer in place)."""
    del _SEGMENTS[:]


def _segments(idx_flat, n_out, stable=True):
    """Pair indices sorted stably by target row (int32) and the (n_out + 1) segment bounds (stable=False: grouped by target
    row in unspecified order inside a group, built by the library's counting sort -- kernels only, so it can be captured;
    torch.sort inside a captured graph is not replayed correctly on this stack).  Cached on the STORAGE the
    indices live in (+ offset, version, length) -- callers hand in a fresh `.view(-1)` object every time, so the tensor
    object's identity says nothing; the storage's owner is kept alive so that its address cannot be recycled.
    While a stream is being captured the cache is neither read nor written: a hit would bake tensors of an EARLIER
    (eager) call into the graph -- constants of the normal memory pool that the cache frees on its next eviction while
    every replay still reads them -- and an entry made during the capture would hand graph-pool tensors to later eager
    calls.
    INVARIANT the cache relies on: an index buffer is never refilled IN PLACE by a library kernel between two
    uses -- the library writes through raw pointers and does not bump `_version`.  Every producer of neighbour lists in
    this package (ops.knn, ops.fps*, gather / nested-level helpers) allocates a fresh output, so it holds; a caller that
    reuses one index buffer for different lists (`out=` style static buffers) must call forget_segments() after
    refilling it."""
    capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    stable = bool(stable) or n_out > 16384
    assert not (capturing and stable), 'the stable (deterministic) segmentation sorts with torch: eager steps only'
    key = (idx_flat.untyped_storage().data_ptr(), idx_flat.storage_offset(), idx_flat._version, idx_flat.numel(),
           idx_flat.stride(0) if idx_flat.numel() > 1 else 1, n_out, stable)
    if not capturing:
        for k, keep, order, off, built in _SEGMENTS:
            if k == key:
                if built is not None:
                    torch.cuda.current_stream().wait_event(built)     # (a hit from another stream than the builder's)
                return order, off
    if stable:
        keys, order = torch.sort(idx_flat.long(), stable=True)
        order = order.to(torch.int32)
        off = torch.searchsorted(keys, torch.arange(n_out + 1, device=idx_flat.device)).to(torch.int32)
    else:
        idx_c = idx_flat if idx_flat.is_contiguous() else idx_flat.contiguous()
        order = torch.empty((idx_c.numel(),), dtype=torch.int32, device=idx_c.device)
        off = torch.empty((n_out + 1,), dtype=torch.int32, device=idx_c.device)
        ws = torch.empty((int(_lib.lib().occ4d_segments_workspace_ints(n_out)),), dtype=torch.int32, device=idx_c.device)
        _lib.check(_lib.lib().occ4d_segments_build_i32(_ptr(idx_c), idx_c.numel(), n_out, _ptr(order), _ptr(off), _ptr(ws),
                                                      _stream()))
    if not capturing:
        built = None
        if idx_flat.is_cuda:
            built = torch.cuda.Event()
            built.record()
        _SEGMENTS.append((key, idx_flat, order, off, built))
        del _SEGMENTS[:-4]
    return order, off


def segment_gather_sum(src, idx_flat, n_out, scale=1.0, weights=None, div=1):
    """out[r] = scale * sum over the pairs p with idx_flat[p] == r, in pair order, of [weights[p] *] src[p // div]."""
    src, lds = _rows(_dev(src, name='src'), 'src')
    order, off = _segments(idx_flat, n_out)
    d = src.shape[1]
    out = torch.empty((n_out, d), dtype=torch.float32, device=src.device)
    wv = None if weights is None else _cont(weights, 'weights').view(-1)
    assert idx_flat.numel() == src.shape[0] * div and (wv is None or wv.numel() == idx_flat.numel())
    _lib.check(_lib.lib().occ4d_segment_gather_sum_f32(_ptr(src), lds, _ptr(order), _ptr(off), _ptr(wv), int(div),
                                                       n_out, d, float(scale), _ptr(out), d, _stream()))
    return out


# Large scatters onto few target rows (the key-table gradient of the attention backward: 458752 pair rows of 832 floats
# onto 4248 abstract points) as a sorted-segment sum instead of fp32 atomics: the pair gradient is read once at HBM speed
# and every output row is written once (the atomics run at 2.7 TB/s: 0.66 T float-atomics / s at the L2).  The sort of one
# neighbour list serves both attention layers (segment cache).  OCC4D_SORTED_SCATTER=0: atomics everywhere.
SORTED_SCATTER = os.environ.get('OCC4D_SORTED_SCATTER', '1') != '0'
SORTED_SCATTER_PARTS = int(os.environ.get('OCC4D_SORTED_SCATTER_PARTS', '8'))
SORTED_SCATTER_RATIO = int(os.environ.get('OCC4D_SORTED_SCATTER_RATIO', '16'))


def scatter_add_rows(src, idx, n_out, scale=1.0):
    src, lds = _rows(_dev(src, name='src'), 'src')
    idx = _dev(idx, torch.int32, 'idx').contiguous().view(-1)
    n, d = src.shape
    assert idx.numel() == n
    if DETERMINISTIC:
        return segment_gather_sum(src, idx, n_out, scale=scale)
    if SORTED_SCATTER and n >= 65536 and SORTED_SCATTER_RATIO * n_out <= n and d >= 64 and d % 4 == 0 and lds % 4 == 0 \
            and src.data_ptr() % 16 == 0 and n_out <= 16384:
        order, off = _segments(idx, n_out, stable=False)
        out = torch.empty((n_out, d), dtype=torch.float32, device=src.device)
        _lib.check(_lib.lib().occ4d_segment_sum_sorted_f32(_ptr(src), lds, _ptr(order), _ptr(off), n_out, d, SORTED_SCATTER_PARTS,
                                                          float(scale), _ptr(out), d, _stream()))
        return out
    out = torch.zeros((n_out, d), dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().occ4d_scatter_add_rows_f32(_ptr(src), lds, _ptr(idx), n, d, float(scale), _ptr(out), d,
                                                     _stream()))
    return out


def segment_sum(src, k):
    src = _cont(src, 'src')
    nk, d = src.shape
    assert nk % k == 0
    out = torch.empty((nk // k, d), dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().occ4d_segment_sum_f32(_ptr(src), nk // k, k, d, _ptr(out), d, _stream()))
    return out


def maxpool_gather_bwd(y, idx, dz):
    y, ldy = _rows(_dev(y, name='y'), 'y')
    dz, ldz = _rows(_dev(dz, name='dz'), 'dz')
    idx = _dev(idx, torch.int32, 'idx')
    n_out, k = idx.shape
    d = y.shape[1]
    if DETERMINISTIC:
        # element-wise scatter (the target row differs per channel): first argmax over the k neighbours, then the
        # (n_out * d) elements of dz summed per target element in a fixed order
        best = y[idx.long()].argmax(dim=1)                                  # (n_out, d): first maximum (torch rule)
        target = (torch.gather(idx.long(), 1, best) * d + torch.arange(d, device=y.device)).view(-1).to(torch.int32)
        flat = segment_gather_sum(dz.contiguous().view(-1, 1), target, y.shape[0] * d)
        return flat.view(y.shape[0], d)
    dy = torch.zeros_like(y, memory_format=torch.contiguous_format)
    _lib.check(_lib.lib().occ4d_maxpool_gather_bwd_f32(_ptr(y), ldy, _ptr(idx), n_out, k, d, _ptr(dz), ldz, _ptr(dy), d,
                                                       _stream()))
    return dy


def layernorm_bwd(x, gamma, g, eps):
    x, ldx = _rows(_dev(x, name='x'), 'x')
    g, ldg = _rows(_dev(g, name='g'), 'g')
    n, d = x.shape
    dx = torch.empty((n, d), dtype=torch.float32, device=x.device)
    if DETERMINISTIC:
        # the kernel gives dx; the two parameter gradients are plain column sums, taken by torch's (fixed-tree) reduction
        _lib.check(_lib.lib().occ4d_layernorm_bwd_f32(_ptr(x), ldx, _ptr(_cont(gamma)), _ptr(g), ldg, float(eps), n, d,
                                                      _ptr(dx), d, None, None, _stream()))
        mu = x.mean(dim=1, keepdim=True)
        xh = (x - mu) * torch.rsqrt(((x - mu) ** 2).mean(dim=1, keepdim=True) + eps)
        return dx, (g * xh).sum(dim=0), g.sum(dim=0)
    dgamma = torch.zeros((d,), dtype=torch.float32, device=x.device)
    dbeta = torch.zeros((d,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().occ4d_layernorm_bwd_f32(_ptr(x), ldx, _ptr(_cont(gamma)), _ptr(g), ldg, float(eps), n, d,
                                                  _ptr(dx), d, _ptr(dgamma), _ptr(dbeta), _stream()))
    return dx, dgamma, dbeta


SOFTMAX_BWD_SPLIT = os.environ.get('OCC4D_SOFTMAX_BWD4', '1') != '0'


def pt_softmax_agg_bwd(logits, v, pe, idx, dagg, reduce_dv=True):
    """(dlogits, dpe, dv).  reduce_dv=False: where the kernel leaves per-pair value gradients, dv comes back as the
    triple (dval, idx32, m) for the caller to reduce (`scatter_add_rows(dval, idx32, m)`), e.g. on another stream."""
    logits = _cont(logits, 'logits')
    v, ldv = _rows(_dev(v, name='v'), 'v')
    dagg, ldda = _rows(_dev(dagg, name='dagg'), 'dagg')
    n, k = idx.shape
    d = logits.shape[1]
    pe = _cont(pe, 'pe') if pe is not None else None
    dlogits = torch.empty_like(logits)
    divisor = float(torch.tensor(math.sqrt(d), dtype=torch.float32))
    idx32 = _dev(idx, torch.int32)
    if DETERMINISTIC:
        # the per-pair value gradients (the kernel's dpe output) are kept and summed per abstract point in pair order
        # (measured in round 4: routing this sum through occ4d_segment_sum_sorted_f32 in the default mode instead of the
        # kernel's own atomics changes nothing, 103.95 vs 103.91 ms per step)
        dval = torch.empty_like(logits)
        _lib.check(_lib.lib().occ4d_pt_softmax_agg_bwd_f32(_ptr(logits), _ptr(v), ldv, _ptr(pe), _ptr(idx32), n, k, d,
                                                           divisor, _ptr(dagg), ldda, _ptr(dlogits), _ptr(dval), None, d,
                                                           _stream()))
        dv = segment_gather_sum(dval, idx32.contiguous().view(-1), v.shape[0])
        return dlogits, (dval if pe is not None else None), dv
    if SOFTMAX_BWD_SPLIT and k in (8, 12, 14, 16) and d % 4 == 0 and ldv % 4 == 0 and ldda % 4 == 0:
        # 16-byte-lane kernel (5.4 TB/s over the four pair tensors) writes the per-pair value gradients; their sum per
        # abstract point is a separate reduction (its own atomics from 16-byte lanes cost more than the kernel)
        dval = torch.empty_like(logits)
        _lib.check(_lib.lib().occ4d_pt_softmax_agg_bwd_f32(_ptr(logits), _ptr(v), ldv, _ptr(pe), _ptr(idx32), n, k, d,
                                                           divisor, _ptr(dagg), ldda, _ptr(dlogits), _ptr(dval), None, d,
                                                           _stream()))
        dv = scatter_add_rows(dval, idx32, v.shape[0]) if reduce_dv else (dval, idx32, v.shape[0])
        return dlogits, (dval if pe is not None else None), dv
    dpe = torch.empty_like(logits) if pe is not None else None
    dv = torch.zeros((v.shape[0], d), dtype=torch.float32, device=logits.device)
    _lib.check(_lib.lib().occ4d_pt_softmax_agg_bwd_f32(_ptr(logits), _ptr(v), ldv, _ptr(pe),
                                                       _ptr(idx32), n, k, d, divisor, _ptr(dagg), ldda,
                                                       _ptr(dlogits), _ptr(dpe), _ptr(dv), d, _stream()))
    return dlogits, dpe, dv


def pt_pos_hidden_bwd(pos, pos2, idx, r, gr):
    p, ps = _rows(_dev(pos, name='pos'), 'pos')
    p2, p2s = _rows(_dev(pos2, name='pos2'), 'pos2')
    n, k = idx.shape
    r = _cont(r, 'r')
    gr = _cont(gr, 'gr')
    h = r.shape[1]
    dP1 = torch.zeros((h, 3), dtype=torch.float32, device=r.device)
    dc1 = torch.zeros((h,), dtype=torch.float32, device=r.device)
    if DETERMINISTIC:
        floats = C.c_int64(0)
        _lib.check(_lib.lib().occ4d_pt_pos_hidden_bwd_det_workspace(n, k, h, C.byref(floats)))
        ws = torch.empty((floats.value,), dtype=torch.float32, device=r.device)
        _lib.check(_lib.lib().occ4d_pt_pos_hidden_bwd_det_f32(_ptr(p), ps, _ptr(p2), p2s, _ptr(_dev(idx, torch.int32)), n, k,
                                                              h, _ptr(r), _ptr(gr), _ptr(dP1), _ptr(dc1), _ptr(ws),
                                                              _stream()))
        return dP1, dc1
    _lib.check(_lib.lib().occ4d_pt_pos_hidden_bwd_f32(_ptr(p), ps, _ptr(p2), p2s, _ptr(_dev(idx, torch.int32)), n, k, h,
                                                      _ptr(r), _ptr(gr), _ptr(dP1), _ptr(dc1), _stream()))
    return dP1, dc1


def interp_bwd(dy, idx, w, n_table):
    dy, ldy = _rows(_dev(dy, name='dy'), 'dy')
    n, k = idx.shape
    d = dy.shape[1]
    if DETERMINISTIC:
        return segment_gather_sum(dy, _dev(idx, torch.int32).contiguous().view(-1), n_table, weights=w, div=k)
    dtable = torch.zeros((n_table, d), dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().occ4d_interp_bwd_f32(_ptr(dy), ldy, _ptr(_dev(idx, torch.int32)), _ptr(_cont(w)), n, k, d,
                                               _ptr(dtable), d, _stream()))
    return dtable


def broadcast_rows(vec, n, scale=1.0):
    vec = _cont(vec, 'vec')
    d = vec.numel()
    out = torch.empty((n, d), dtype=torch.float32, device=vec.device)
    _lib.check(_lib.lib().occ4d_broadcast_rows_f32(_ptr(vec), float(scale), n, d, _ptr(out), d, _stream()))
    return out
