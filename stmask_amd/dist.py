"""Frame-level data parallelism over independent video clips: one process per GPU, clip i -> rank i mod world, and ONE
fixed-shape all-gather of detections per step (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

Payload per clip and step: [top_k, 40] fp32 = (box 4, score, class, id, valid, coeff 32) = 32 KB -- latency bound, far
below one xGMI link, so a single direct all_gather is the right collective (no ring / bucket tuning applies).  Nothing
in the reference corresponds to this: it has no inference-time multi-GPU path.
"""
import torch
import torch.distributed as dist

DET_COLS = 40  # 4 box + score + class + box_id + valid + 32 mask coefficients


def shard_clips(n_clips, rank, world):
    """Indices of the clips this rank owns (clip i -> rank i mod world)."""
    return list(range(rank, n_clips, world))


def pack_detections(dets, top_k=200, device=None):
    """list of detection dicts -> [len(dets), top_k, DET_COLS] fp32, zero padded, column 7 = valid flag."""
    device = device or (dets[0]["box"].device if dets else "cpu")
    out = torch.zeros(len(dets), top_k, DET_COLS, device=device)
    for i, d in enumerate(dets):
        n = min(d["box"].shape[0], top_k) if d["box"].numel() else 0
        if n == 0:
            continue
        out[i, :n, 0:4] = d["box"][:n]
        out[i, :n, 4] = d["score"][:n]
        out[i, :n, 5] = d["class"][:n].float()
        out[i, :n, 6] = d["box_ids"][:n].float()
        out[i, :n, 7] = 1.0
        out[i, :n, 8:8 + d["mask_coeff"].shape[1]] = d["mask_coeff"][:n]
    return out


def unpack_detections(packed):
    """Inverse of pack_detections for one clip: [top_k, DET_COLS] -> dict of the valid rows."""
    valid = packed[:, 7] > 0
    p = packed[valid]
    return {"box": p[:, 0:4], "score": p[:, 4], "class": p[:, 5].long(), "box_ids": p[:, 6].long(), "mask_coeff": p[:, 8:]}


def all_gather_detections(packed):
    """[local_clips, top_k, DET_COLS] on every rank -> [world * local_clips, top_k, DET_COLS] (rank-major).
    Every rank must pass the same local_clips (pad the last shard).  Blocking form (the current stream waits for the
    collective); the benchmark's steady state uses DetectionGatherer, which keeps the collective off the compute streams."""
    if not (dist.is_available() and dist.is_initialized()):
        return packed
    world = dist.get_world_size()
    if packed.is_cuda and dist.get_backend() == "gloo":
        # gloo moves host memory: stage through the host (the two-ranks-on-one-GPU check of bench.py, CPU tests)
        host = packed.contiguous().cpu()
        out = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host)
        return out.to(packed.device, non_blocking=True)
    out = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed.contiguous())
    return out


class DetectionGatherer:
    """The per-step all-gather of SURVEY.md section 8(e), kept off the compute streams: the packed detections of step t are
    gathered on a communication stream of their own (it waits for the stream that produced them; RCCL's collective then
    synchronises with THAT stream), while the main stream goes straight on to step t + 1 and the side stream keeps running the
    next trunk.  Nothing on the device consumes the gathered rows (they feed the host-side result aggregation), so no compute
    stream ever waits for the 32 KB-per-clip exchange.  `gather()` returns the gathered tensor; it is complete once `wait()`
    (or a device synchronisation) has returned."""

    def __init__(self, device=None):
        self.device = device
        self._comm = None
        self._events = []
        self.n_collectives = 0          # all-gathers enqueued on the communication stream (the RCCL branch)
        # diagnosis forms (an environment switch in round 5, profiles/r05_launcher_overhead.txt; bench.py sets 2 for its no-gather pass): 0 = as described above; 1 = two persistent output buffers, the packed block kept alive by a
        # reference instead of record_stream; 2 = no collective at all (the process group exists, nothing is gathered); 3 = the collective on the
        # CURRENT stream (no communication stream)
        self.mode = 0
        self._outs, self._hold = [None, None], None

    def gather(self, packed):
        if not (dist.is_available() and dist.is_initialized()):
            return packed
        if not packed.is_cuda or dist.get_backend() == "gloo":
            return all_gather_detections(packed)
        if self.mode == 2:
            return packed
        if self.mode == 3:
            self.n_collectives += 1
            return all_gather_detections(packed)
        self.comm_stream(packed.device)
        main = torch.cuda.current_stream()
        self._comm.wait_stream(main)
        with torch.cuda.stream(self._comm):
            if self.mode == 1:
                i = self.n_collectives & 1
                shape = (dist.get_world_size() * packed.shape[0],) + tuple(packed.shape[1:])
                if self._outs[i] is None or tuple(self._outs[i].shape) != shape:
                    self._outs[i] = torch.empty(shape, dtype=packed.dtype, device=packed.device)
                out = self._outs[i]
                self._hold = packed.contiguous()
                dist.all_gather_into_tensor(out, self._hold)
            else:
                packed.record_stream(self._comm)
                out = all_gather_detections(packed)
            ev = torch.cuda.Event()
            ev.record()
        self._events = [ev]
        self.n_collectives += 1
        return out

    def comm_stream(self, device):
        """The communication stream: one that runs beside the main stream AND beside every trunk stream of the pipeline (streams that share a
        hardware queue serialise: pipeline.concurrent_side_streams) -- the exchange must not queue up behind a prefetched trunk.  The pipelines
        rotate their trunks over the first trunk_stream_count() streams of that list; the gather takes the one after them."""
        if self._comm is None:
            from .pipeline import concurrent_side_streams, trunk_stream_count
            self._comm = concurrent_side_streams(device, trunk_stream_count() + 1)[-1]
        return self._comm

    def wait(self):
        """Host-side: the last gather() has completed."""
        for ev in self._events:
            ev.synchronize()
        self._events = []


def global_clip_order(n_clips, world):
    """Position in the rank-major gathered tensor of global clip c (for un-sharding): clip c lives on rank c % world at
    local slot c // world."""
    per = (n_clips + world - 1) // world
    return [(c % world) * per + c // world for c in range(n_clips)]
