"""CPU, world_size 2, gloo: the query-sharding / broadcast / gather logic of
occlusions-4d_amd/distributed.py.  The HIP modules cannot run without a GPU, so the two
networks are stand-ins that expose the same interface and compute with the CPU oracle; what
is under test is the host-side distribution logic (who encodes, what is broadcast, which
slice each rank decodes, how outputs are gathered), checked against the single-process
oracle result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import occlusions4d_amd as pk
from oracle import path as op


class OracleEncoder(torch.nn.Module):
    def __init__(self, sd, pa, calls):
        super().__init__()
        self.sd, self.pa, self.calls = sd, pa, calls
        self.down_blocks, self.transition_factor = pa['down_blocks'], pa['transition_factor']
        self.abstract_levels, self.d_feat, self.global_dim = pa['abstract_levels'], pa['d_feat'], pa['global_dim']

    def forward(self, pcl, return_intermediate):
        self.calls.append('encode')
        out, xg = op.encoder_forward(self.sd, self.pa, pcl)
        return out, xg, None


class OracleDecoder(torch.nn.Module):
    def __init__(self, sd, ia):
        super().__init__()
        self.sd, self.ia, self.d_out = sd, ia, ia['d_out']

    def forward(self, q, abstract, fglob, _):
        return op.decoder_forward(self.sd, self.ia, q, abstract, fglob)


def cpu_squash(out, codes):
    for c, code in enumerate(codes):
        if code == 1:
            out[:, c] = torch.sigmoid(out[:, c])
        elif code == 2:
            out[:, c] = torch.clamp(out[:, c], 0.0, 1.0)
    return out


def _worker(rank, world, port, kind, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        case = dict(kind=kind, n=384, video_len=4, seed=77)
        pa, ia, inf = pk.configs.model_args(kind, case['n'])
        pcl = pk.configs.synthetic_pcl(kind, case['n'], 4, 77)
        esd, dsd = pk.configs.synthetic_weights(pa, ia, 77)
        calls = []
        enc, dec = OracleEncoder(esd, pa, calls), OracleDecoder(dsd, ia)
        q = torch.from_numpy(op.sample_query_points(700, inf['min_z'], inf['cube_bounds'], 1, kind, 4, 'grid'))
        with torch.no_grad():
            full = pk.distributed.sharded_inference(pcl, q, enc, dec, 128, inf['color_mode'],
                                                    inf['predict_segmentation'], 'none', 13, gather=True,
                                                    squash=cpu_squash)
            local, (lo, hi) = pk.distributed.sharded_inference(pcl, q, enc, dec, 128, inf['color_mode'],
                                                               inf['predict_segmentation'], 'none', 13,
                                                               squash=cpu_squash)
        assert torch.equal(full[lo:hi], local)
        assert calls == (['encode', 'encode'] if rank == 0 else []), calls   # only rank 0 encodes
        # ONE packed broadcast per clip (round 6): the abstract cloud and the global embedding are views of one buffer
        sent = []
        real_broadcast = dist.broadcast

        def counting_broadcast(tensor, src=0, **kw):
            sent.append(tuple(tensor.shape))
            return real_broadcast(tensor, src=src, **kw)
        dist.broadcast = counting_broadcast
        try:
            shape = pk.distributed.abstract_shape(enc, case['n'])
            ab, fg = pk.distributed.encode_and_share(pcl, enc, shape, enc.global_dim, torch.device('cpu'))
        finally:
            dist.broadcast = real_broadcast
        off, total = pk.distributed.packed_layout(shape, enc.global_dim)
        assert sent == [(total,)] and off % 4 == 0 and total == off + enc.global_dim
        assert ab.untyped_storage().data_ptr() == fg.untyped_storage().data_ptr() and ab.is_contiguous()
        ref_ab, ref_fg = op.encoder_forward(esd, pa, pcl)
        assert torch.equal(ab, ref_ab[0]) and torch.equal(fg, ref_fg[0])
        # the pipelined schedule (ClipPipeline: encode + broadcast of clip i + 1 issued before clip i decodes) over a
        # stream of DIFFERENT clips = the sequential schedule, bit for bit, on every rank
        clips = [pk.configs.synthetic_pcl(kind, case['n'], 4, 90 + i) for i in range(3)]
        with torch.no_grad():
            seq = [pk.distributed.sharded_inference(c, q, enc, dec, 128, inf['color_mode'], inf['predict_segmentation'],
                                                    'none', 13, gather=True, squash=cpu_squash) for c in clips]
            del calls[:]
            pipe = pk.distributed.ClipPipeline(enc, dec, 128, inf['color_mode'], inf['predict_segmentation'], 'none', 13,
                                               squash=cpu_squash)
            piped = list(pipe.run(clips, q, gather=True))
        assert len(piped) == 3 and all(torch.equal(a, b) for a, b in zip(seq, piped))
        assert calls == (['encode'] * 3 if rank == 0 else []), calls
        if rank == 0:
            ret['full'] = full.numpy()
            ret['n'] = q.shape[0]
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('kind', ['greater', 'carla'])
def test_sharded_inference_two_ranks_matches_single_process(kind):
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0
        full = ret['full']
    pa, ia, inf = pk.configs.model_args(kind, 384)
    pcl = pk.configs.synthetic_pcl(kind, 384, 4, 77)
    esd, dsd = pk.configs.synthetic_weights(pa, ia, 77)
    ref = op.perform_inference(pcl, esd, pa, dsd, ia, inf['min_z'], inf['cube_bounds'], inf['color_mode'], 1,
                               num_sample=700, point_sample_mode='grid', batch_size=128,
                               predict_segmentation=inf['predict_segmentation'], track_mode='none',
                               semantic_classes=13, data_kind=kind, cube_mode=4)
    assert full.shape == ref['implicit_output'].shape
    # identical arithmetic on both sides up to batch-boundary placement (row-independent ops)
    np.testing.assert_allclose(full, ref['implicit_output'], rtol=0, atol=2e-6)


def test_shard_bounds_cover_exactly_once():
    for n in (0, 1, 7, 534528, 2125568):
        for world in (1, 2, 3, 4, 8):
            spans = [pk.distributed.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(lo <= hi for lo, hi in spans)


def test_abstract_shape_matches_survey_sizes():
    class E:
        down_blocks, transition_factor, d_feat = 3, 3, 36
    E.abstract_levels = 1
    assert pk.distributed.abstract_shape(E, 14336) == (531, 291)
    assert pk.distributed.abstract_shape(E, 2048) == (76, 291)
    E.abstract_levels = 2
    assert pk.distributed.abstract_shape(E, 14336) == (2124, 291)
    assert pk.distributed.abstract_shape(E, 28672) == (4248, 291)


def _grad_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)),
                  torch.nn.Parameter(torch.zeros(2, 2))]
        params[0].grad = torch.full((5, 3), float(rank + 1))
        params[1].grad = torch.arange(7, dtype=torch.float32) * (rank + 1)
        # params[2] has a gradient on rank 1 only: rank 0 takes part with zeros (every rank reduces the same flat
        # layout).  params[3] has a gradient on NO rank: it must keep grad = None, so that the optimiser skips it as in a
        # single-process run (no weight decay / moment update for an unused parameter)
        params.append(torch.nn.Parameter(torch.zeros(3)))
        if rank == 1:
            params[2].grad = torch.full((2, 2), 4.0)
        part = pk.training.Participation()
        pk.training.allreduce_gradients(params, participation=part)
        assert part.used == [True, True, True, False]       # the mask a captured step would reuse lives on the caller
        ret[rank] = [None if p.grad is None else p.grad.clone() for p in params]
        # round 5: a second step REUSES the agreement (no mask all-reduce, no host read): rank 0 comes without a gradient
        # for params[2] again and still reduces it as zeros; the unused parameter stays untouched
        agreed = part.used
        params[0].grad = torch.full((5, 3), 10.0 * (rank + 1))
        if rank == 0:
            params[2].grad = None
        else:
            params[2].grad = torch.full((2, 2), 8.0)
        calls = []
        orig = dist.all_reduce
        dist.all_reduce = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            pk.training.allreduce_gradients(params, participation=part)
        finally:
            dist.all_reduce = orig
        assert len(calls) == 1 and part.used is agreed       # the flat bucket only
        assert torch.allclose(params[0].grad, torch.full((5, 3), 15.0)) and torch.allclose(params[2].grad, torch.full((2, 2), 4.0))
        assert params[3].grad is None
        # a gradient that turns up for a parameter the ranks agreed nobody uses is an error, not a silent skip
        params[3].grad = torch.ones(3)
        try:
            pk.training.allreduce_gradients(params, participation=part)
            raise SystemExit('expected a RuntimeError')
        except RuntimeError as e:
            assert 'Participation.clear()' in str(e)
        params[3].grad = None
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_averages_over_ranks():
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        out = {r: ret[r] for r in range(2)}
    for r in range(2):
        assert torch.allclose(out[r][0], torch.full((5, 3), 1.5))
        assert torch.allclose(out[r][1], torch.arange(7, dtype=torch.float32) * 1.5)
        assert torch.allclose(out[r][2], torch.full((2, 2), 2.0))
        assert out[r][3] is None
