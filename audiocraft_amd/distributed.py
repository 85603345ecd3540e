"""Batched-prompt sharding across the GPUs of one node (RCCL over xGMI through torch.distributed).

The reference has no multi-GPU inference path; each prompt's generation is
independent, so the path shards with NO collective inside the autoregressive loop:

  1. rank 0 builds the condition tensors for the GLOBAL batch (so the padded text length Lc -- which
     changes every prompt's logits because padding is not masked in cross-attention, SURVEY.md
     section 7 -- is the same as on a single device) and BROADCASTS them           (1 collective)
  2. every rank keeps its rows of the `[cond; uncond]` batch and generates + decodes locally
  3. tokens (48 KB / sample) are ALL-GATHERED so every rank (and the caller on rank 0) sees the
     full [B, K, T] result                                                          (1 collective)

Both messages are far below one xGMI link's bandwidth; the scaling is weight-replicated "weak".
Works with any torch.distributed backend ("nccl" == RCCL on ROCm; "gloo" in the CPU tests).
"""
import os
import typing as tp

import torch
import torch.distributed as dist

ConditionTensors = tp.Dict[str, tp.Tuple[torch.Tensor, torch.Tensor]]


def shared_device_allowed() -> bool:
    """ACMI_ALLOW_SHARED_DEVICE=1: several ranks may drive ONE device (rank r -> device r % device_count).  A test switch:
    it lets a 1-GPU box execute the whole multi-process path (spawn, rendezvous, broadcast, sharded generate with the real
    kernels, all-gather, the max-over-ranks clock).  RCCL refuses two ranks on one device, so it goes with
    ACMI_DIST_BACKEND=gloo; production stays one process per GPU over RCCL."""
    return os.environ.get('ACMI_ALLOW_SHARED_DEVICE', '') == '1'


def device_index(local_rank: int) -> int:
    """HIP device of this rank: LOCAL_RANK (one process per GPU), or LOCAL_RANK % device_count with the shared-device switch."""
    n_dev = torch.cuda.device_count()
    if shared_device_allowed() and n_dev > 0:
        return local_rank % n_dev
    if local_rank >= n_dev:   # one process per GPU: LOCAL_RANK r drives HIP device r of this node
        raise RuntimeError(f"LOCAL_RANK={local_rank} but this node shows {n_dev} device(s) "
                           f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')})")
    return local_rank


def require_own_device():
    """The generation path runs on a device: a rank that was started beyond the node's devices (a host-side transport lets it
    through the rendezvous) must not silently share device 0 with another rank -- that is device sharing without
    ACMI_ALLOW_SHARED_DEVICE and skews the max-over-ranks clock.  Raises with the remedy; no-op for a single process."""
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1 or not torch.cuda.is_available():
        return
    device_index(int(os.environ.get('LOCAL_RANK', '0')))   # raises "LOCAL_RANK=r but this node shows n device(s)" unless the switch is on


def init_from_env(backend: tp.Optional[str] = None) -> tp.Tuple[int, int, int]:
    """-> (rank, world_size, local_rank); initialises the default process group when WORLD_SIZE > 1.
    Backend: the argument, else ACMI_DIST_BACKEND, else "nccl" (= RCCL) with a GPU and "gloo" without."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('ACMI_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            if shared_device_allowed() and world > torch.cuda.device_count():
                raise RuntimeError("ACMI_ALLOW_SHARED_DEVICE=1 needs ACMI_DIST_BACKEND=gloo: RCCL refuses two ranks on one device")
            idx = device_index(local_rank)
            torch.cuda.set_device(idx)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', idx))
        else:
            # a host-side transport: bind a device only when this rank has one (one per rank, or the shared-device
            # switch); a rank beyond the node's devices stays on the host instead of failing before the rendezvous.
            # The strict LOCAL_RANK < device_count check belongs to the RCCL branch only.
            if torch.cuda.is_available() and (shared_device_allowed() or local_rank < torch.cuda.device_count()):
                torch.cuda.set_device(device_index(local_rank))
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def _staged(t: torch.Tensor) -> torch.Tensor:
    """The tensor a collective runs on: gloo moves host memory only (its device support covers broadcast / all_reduce, not
    all_gather), so with that backend device tensors are staged through the host; RCCL takes them as they are."""
    if not t.is_cuda:
        return t
    try:
        gloo = dist.get_backend() == 'gloo'
    except (ValueError, RuntimeError):   # no default group (a caller-provided transport): nothing to stage for
        gloo = False
    return t.cpu() if gloo else t


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def shard_range(total: int, rk: int, world: int) -> tp.Tuple[int, int]:
    """Contiguous shard of `total` items for rank `rk`; the first (total % world) ranks get one more."""
    base, extra = divmod(total, world)
    start = rk * base + min(rk, extra)
    return start, start + base + (1 if rk < extra else 0)


_MAX_CONDS, _NAME_BYTES = 8, 32   # header layout: per condition 4 int64 words of name + (rows, L, d)


def _pack_header(ct: ConditionTensors) -> torch.Tensor:
    assert len(ct) <= _MAX_CONDS, f"at most {_MAX_CONDS} conditions"
    h = torch.zeros(_MAX_CONDS, _NAME_BYTES // 8 + 3, dtype=torch.int64)
    for i, (name, (e, m)) in enumerate(ct.items()):
        raw = name.encode()
        assert 0 < len(raw) <= _NAME_BYTES, f"condition name '{name}' longer than {_NAME_BYTES} bytes"
        h[i, :_NAME_BYTES // 8] = torch.frombuffer(bytearray(raw.ljust(_NAME_BYTES, b'\0')), dtype=torch.int64)
        assert e.dim() == 3 and tuple(m.shape) == tuple(e.shape[:2]), (name, tuple(e.shape), tuple(m.shape))
        h[i, _NAME_BYTES // 8:] = torch.tensor(e.shape, dtype=torch.int64)
    return h


def _unpack_header(h: torch.Tensor) -> tp.List[tp.Tuple[str, tp.Tuple[int, int, int]]]:
    out = []
    for row in h.cpu():
        rows, L, d = (int(v) for v in row[_NAME_BYTES // 8:])
        if rows == 0:
            continue
        name = row[:_NAME_BYTES // 8].contiguous().numpy().tobytes().rstrip(b'\0').decode()
        out.append((name, (rows, L, d)))
    return out


def broadcast_condition_tensors(ct: tp.Optional[ConditionTensors], device, src: int = 0) -> ConditionTensors:
    """Broadcast {name: (emb [rows, L, d] f32, mask [rows, L] int64)} from `src`; other ranks pass None.
    TWO collectives whatever the number of conditions: a fixed-layout int64 header (names, shapes) and one f32 payload
    (embeddings and masks back to back) -- no pickling (`broadcast_object_list`) on the path."""
    if not dist.is_initialized():      # (a process group of ONE rank still runs its collectives: the 1-GPU RCCL test)
        assert ct is not None
        return ct
    if rank() == src:
        assert ct is not None
        header = _pack_header(ct).to(device)
    else:
        header = torch.zeros(_MAX_CONDS, _NAME_BYTES // 8 + 3, dtype=torch.int64, device=device)
    hs = _staged(header)
    dist.broadcast(hs, src=src)
    header = hs.to(header.device)
    layout = _unpack_header(header)
    total = sum(rows * L * d + rows * L for _, (rows, L, d) in layout)
    if rank() == src:
        payload = torch.cat([t for name, _ in layout for t in (
            ct[name][0].to(device=device, dtype=torch.float32).reshape(-1),
            ct[name][1].to(device=device, dtype=torch.float32).reshape(-1))])
        assert payload.numel() == total
    else:
        payload = torch.empty(total, device=device, dtype=torch.float32)
    ps = _staged(payload)
    dist.broadcast(ps, src=src)
    payload = ps.to(payload.device)
    out: ConditionTensors = {}
    off = 0
    for name, (rows, L, d) in layout:
        e = payload[off:off + rows * L * d].view(rows, L, d)
        off += rows * L * d
        m = payload[off:off + rows * L].view(rows, L).to(torch.int64)
        off += rows * L
        out[name] = (e.contiguous(), m.contiguous())
    return out


def shard_condition_tensors(ct: ConditionTensors, B_global: int, rk: int, world: int) -> ConditionTensors:
    """Rows of the `[cond(0..B-1); uncond(0..B-1)]` batch that belong to this rank's prompts."""
    lo, hi = shard_range(B_global, rk, world)
    out: ConditionTensors = {}
    for name, (e, m) in ct.items():
        assert e.shape[0] == 2 * B_global, f"{name}: expected {2 * B_global} rows, got {e.shape[0]}"
        out[name] = (torch.cat([e[lo:hi], e[B_global + lo:B_global + hi]]).contiguous(),
                     torch.cat([m[lo:hi], m[B_global + lo:B_global + hi]]).contiguous())
    return out


def gather_rows(local: torch.Tensor, B_global: int) -> torch.Tensor:
    """All-gather per-prompt rows [B_local, ...] -> [B_global, ...] in prompt order (uneven shards padded)."""
    world = world_size()
    if not dist.is_initialized():
        return local
    rk = rank()
    max_rows = -(-B_global // world)
    pad = torch.zeros((max_rows,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    pad[:local.shape[0]] = local
    send = _staged(pad.contiguous())
    bufs = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(bufs, send)
    parts = []
    for r in range(world):
        lo, hi = shard_range(B_global, r, world)
        parts.append(bufs[r][:hi - lo].to(local.device))
    del rk
    return torch.cat(parts, dim=0)


def generate_sharded(model, descriptions: tp.Optional[tp.Sequence[tp.Optional[str]]], B_global: int,
                     max_gen_len: int, decode: bool = True, gather_audio: bool = False, base_seed: int = 0,
                     generation_params: tp.Optional[dict] = None):
    """Text-to-music for a global batch sharded over the process group.

    Every rank calls this with the same arguments (`descriptions` is only read on rank 0).
    Returns (tokens [B_global, K, T] on every rank, wav for the local shard or, with gather_audio, global).
    Sampling seeds are per rank (base_seed + rank, cf. reference utils/utils.py:203-223)."""
    rk, world = rank(), world_size()
    require_own_device()
    device = model.device
    # every rank needs at least one prompt: a rank with an empty shard would skip generate() and leave the others
    # blocked in the all-gather
    assert B_global >= world, f"global batch {B_global} < world size {world}: launch fewer ranks"
    ct = None
    if rk == 0:
        assert descriptions is not None and len(descriptions) == B_global
        attributes, _ = model._prepare_tokens_and_attributes(descriptions, None)
        ct = model.lm._cfg_condition_tensors(attributes)
    ct = broadcast_condition_tensors(ct, device)
    local_ct = shard_condition_tensors(ct, B_global, rk, world)
    lo, hi = shard_range(B_global, rk, world)
    params = dict(model.generation_params if generation_params is None else generation_params)
    tokens = model.lm.generate(None, [], num_samples=hi - lo, max_gen_len=max_gen_len, condition_tensors=local_ct,
                               seed=base_seed + rk, **params)
    wav = model.generate_audio(tokens) if decode else None
    all_tokens = gather_rows(tokens, B_global)
    if decode and gather_audio:
        wav = gather_rows(wav, B_global)
    return all_tokens, wav
