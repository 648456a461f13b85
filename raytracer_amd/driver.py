"""Multi-GPU render driver: the reference's four render loops (src/main.cpp:14-158), sharded.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  There is no exchange during the trace: finished pictures are gathered to rank 0, which
alone accumulates motion blur (in sub-frame order, so the rounding is the reference's Picture::mulAdd
order) and writes the PNG/APNG.  The one real exchange step is the photon pass of a STILL: all ranks
work on the same frame, so each emits a slice of the root photons and the photon textures are summed
with one all-reduce (≤ 7 MB) before the bands are traced; frames of an animation keep their photon
pass whole on the rank that renders the frame.

  stills                 8-row bands of the image, interleaved over the ranks (per-pixel cost is very
                         uneven on Julia scenes; the reference hands out single rows dynamically,
                         src/raytracer.cpp:48-53); with RT_SPLIT_PHOTONS=1 the photon pass is split over the
                         ranks + all-reduce (equal to rounding only, hence opt-in)
  still + motion blur    sub-frames round-robin over the ranks
  animation              frames round-robin over the ranks, RGBA8 frames gathered
  animation + blur       (frame, sub-frame) pairs round-robin over the ranks

The renderer is injected: HipRenderer (below) is the product path; the CPU tests inject a stand-in so
that the sharding and gather logic can be checked without a GPU.
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import capi

BAND_ROWS = 8
# HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (4 by default) and two streams on one queue run
# one after the other.  render_units keeps three units in flight on three streams beside the caller's own, a download
# stream and RCCL's: with four queues two of its lanes can land on the same one (measured on the 160-unit C5 file inside
# bench.py's `extra`: 1 510 units/s instead of 1 790).  Eight are asked for when this module is imported before the HIP
# runtime starts and the process has not asked for more; a host that must keep four sets RT_KEEP_HW_QUEUES=1.
HW_QUEUES = 8
if os.environ.get("RT_KEEP_HW_QUEUES", "0") in ("", "0") and not torch.cuda.is_initialized():
    try:
        if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < HW_QUEUES:
            os.environ["GPU_MAX_HW_QUEUES"] = str(HW_QUEUES)
    except ValueError:
        pass
# <caustic steps> from which a still's photon pass is split over the ranks: 1000 steps = 500 000 directions
# per light (0.6 ms on one MI355X at C5's size, against some 50 us for the all-reduce of the textures)
SPLIT_PHOTON_MIN_STEPS = 1000.0


class _DeviceFloats:
    """Device memory owned by the C library, described for torch.as_tensor (no copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class HipRenderer:
    """Renders through the C ABI on one GPU; pictures stay in HBM as torch tensors."""

    def __init__(self, device_index=0):
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.ctx = capi.Context(device_index)
        self.busy_events = None      # a list: render() then brackets its GPU work with event pairs (bench.py)

    def busy_seconds(self):
        """GPU time of the render() calls (this renderer's and its twins': render_units) since busy_events was set to a
        list — the union of their intervals, which overlap when two units are in flight; clears the list."""
        torch.cuda.synchronize()
        pairs = list(self.busy_events)
        for twin in getattr(self, "_twins", []):
            if twin.busy_events:
                pairs += twin.busy_events
                twin.busy_events = []
        self.busy_events = []
        if not pairs:
            return 0.0
        base = pairs[0][0]
        spans = sorted((base.elapsed_time(a), base.elapsed_time(b)) for a, b in pairs)
        total, end = 0.0, None
        for lo, hi in spans:
            if end is None or lo > end:
                total += hi - lo
                end = hi
            elif hi > end:
                total += hi - end
                end = hi
        return total * 1e-3

    def lanes(self):
        """[(renderer, stream)] that render_units alternates its units over: this renderer and its twins (further contexts on
        the same GPU; RT_DRIVER_LANES - 1 of them, two by default) on side streams, so that the photon pass of one unit runs
        beside the trace kernels of the ones before — a C5 frame is a chain of dependent queries that leaves more than half
        of the GPU idle on its own.  Measured on the 160-unit C5 file [MI355X]: 1 390 units/s with two lanes, 1 790 with
        three, 1 530-1 570 with four, 1 440-1 680 with five (the VALU is busy 59 % of a lone C5 render: a third render in
        flight has nothing left to fill; tools/lanes_sweep.sh).  RT_DRIVER_OVERLAP=0: this renderer alone on the current stream."""
        if os.environ.get("RT_DRIVER_OVERLAP", "1") in ("", "0"):
            return [(self, None)]
        n = max(2, int(os.environ.get("RT_DRIVER_LANES", "3")))
        if len(getattr(self, "_twins", [])) != n - 1:
            self.close_twins()
            self._twins = [HipRenderer(self.device.index) for _ in range(n - 1)]
            self._streams = [torch.cuda.Stream(self.device) for _ in range(n)]
        for t in self._twins:
            t.busy_events = [] if self.busy_events is not None else None
            # a twin renders alternate units of the same job: it must run with the options set on this renderer's context
            # (photon order, lens seed, BVH threshold, shadow elision ...), whenever they were set
            if t.ctx.options != self.ctx.options:
                for option in set(t.ctx.options) - set(self.ctx.options):
                    raise RuntimeError("option %d was set on a twin context only" % option)
                for option, value in self.ctx.options.items():
                    if t.ctx.options.get(option) != value:
                        t.ctx.set_option(option, value)
        return [(self, self._streams[0])] + [(t, self._streams[k + 1]) for k, t in enumerate(self._twins)]

    def close_twins(self):
        for t in getattr(self, "_twins", []):
            t.close()
        self._twins = []

    def close(self):
        """Destroys the context (and those of the twins lanes() made)."""
        self.close_twins()
        self.ctx.close()

    def _photon_pass(self, enqueue, defer_check=False):
        """Enqueues a photon pass; if it outgrew the record buffer of the ordered sum (the library has resized it
        by then) the pass runs once more, so that the textures are the reference's bit for bit.  defer_check: the
        caller asks photon_pass_ok() later (asking waits for the pass) and renders again if need be."""
        stream = torch.cuda.current_stream().cuda_stream
        enqueue(stream)
        self._unchecked_pass = defer_check and self.ctx.ordered_photons
        if not defer_check and self.ctx.ordered_photons and not self.ctx.photon_ordered():
            enqueue(stream)

    def photon_pass_ok(self):
        """False if the last render(..., defer_check=True) has to be made again (its photon pass outgrew the record buffer)."""
        unchecked, self._unchecked_pass = getattr(self, "_unchecked_pass", False), False
        return not unchecked or self.ctx.photon_ordered()

    def photon_pass(self, scene, part, n_parts):
        """Uploads the scene and emits slice `part` of `n_parts` of its photon pass.  Returns the photon
        textures as a flat float tensor ALIASING the context's device buffer (None without receivers):
        the caller sums it over the ranks in place, then calls render(..., photons_ready=True)."""
        self.ctx.upload(scene)
        self._photon_pass(lambda stream: self.ctx.photon_generate_part_device(part, n_parts, stream))
        ptr, n = self.ctx.photon_atlas_device()
        if n == 0:
            return None
        return torch.as_tensor(_DeviceFloats(ptr, n), device=self.device)

    def render(self, scene, bands=None, photons_ready=False, defer_check=False):
        stream = torch.cuda.current_stream().cuda_stream
        self._unchecked_pass = False
        if self.busy_events is not None:
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
        if not photons_ready:      # (an upload clears the photon textures)
            self.ctx.upload(scene)
            if scene.desc.photon_steps > 0:
                self._photon_pass(self.ctx.photon_generate_device, defer_check)
        n = self.ctx.pixel_count(bands)
        out = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        if n:
            self.ctx.render_device(out.data_ptr(), stream, bands)
            if scene.desc.n_julias > 0:
                self.ctx.launch_status()     # an abandoned Julia launch must not reach the gather as a picture
        if self.busy_events is not None:
            pair[1].record()
            self.busy_events.append(pair)
        return out

    def mul_add(self, acc, src, factor):
        self.ctx.mul_add_device(acc.data_ptr(), src.data_ptr(), factor, acc.numel(), torch.cuda.current_stream().cuda_stream)

    def quantize(self, rgba):
        out = torch.empty(rgba.shape, dtype=torch.uint8, device=rgba.device)
        self.ctx.quantize_device(rgba.data_ptr(), out.data_ptr(), 1.0, rgba.numel() // 4, torch.cuda.current_stream().cuda_stream)
        return out


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def settle_for_gloo(tensor):
    """Rehearsal only (RT_DIST_BACKEND=gloo with device tensors).  torch 2.10's gloo backend on ROCm 7.2 hangs in
    Work.wait() when a DEVICE tensor is handed to an asynchronous collective while kernels are still in flight on the
    current stream and the host goes on submitting: its worker thread never returns from waiting for the staging copy.
    tools/gloo_cuda_gather_probe.py shows it with torch kernels alone (`torch-long`), nothing of this package loaded;
    with the device idle at the call it does not happen.  RCCL, the product path, has no such thread and is not touched."""
    if tensor.is_cuda and dist.is_initialized() and dist.get_backend() == "gloo":
        torch.cuda.synchronize(tensor.device)


def gather_to_root(tensor):
    """Returns the list of every rank's tensor on rank 0 (None elsewhere).  Equal shapes required."""
    rank, world = world_info()
    if world == 1:
        return [tensor]
    out = [torch.empty_like(tensor) for _ in range(world)] if rank == 0 else None
    settle_for_gloo(tensor)
    dist.gather(tensor, out, dst=0)
    return out


# ---------------------------------------------------------------------------------- schedules

def band_rows_of(height, first_band, stride, band_rows=BAND_ROWS):
    """Image rows rendered by rt_bands{band_rows, first_band, stride}, in output order."""
    rows = []
    for band in range(first_band, (height + band_rows - 1) // band_rows, stride):
        rows.extend(range(band * band_rows, min(height, (band + 1) * band_rows)))
    return rows


def f32(x):
    return np.float32(x)


def still_time(info):
    return f32(0.0) if info.still_time == f32(capi.FLT_MAX) else f32(info.still_time)


def motion_blur_units(path, info, load=capi.Scene):
    """[(frame, sub, t, weight, k)] exactly as the reference's loops produce them, including a
    sub-frame count that the scene file animates (applied from the next frame,
    src/main.cpp:125-151).  Needs only the host loader."""
    units = []
    frames = info.frames
    k = info.sub_frames
    if frames > 1 and info.still_time == f32(capi.FLT_MAX):
        for frame in range(frames):
            next_k = k
            for sub in range(k):
                t = (f32(frame) + f32(sub) / f32(k - 1)) / f32(frames)
                units.append((frame, sub, float(t), float(f32(1.0) / f32(k)), k))
                next_k = load(path, float(t)).info.sub_frames
            k = next_k
    else:
        start = still_time(info)
        k = load(path, float(start)).info.sub_frames
        for sub in range(k):
            t = f32(sub) / f32(k - 1) / f32(frames) + start
            units.append((0, sub, float(t), float(f32(1.0) / f32(k)), k))
    return units


# ---------------------------------------------------------------------------------- the loops

def split_photon_pass(scene, world):
    """Whether the ranks share one photon pass (stills only).  Opt-in (RT_SPLIT_PHOTONS=1): the sum of the ranks'
    partial textures differs from the reference's sequential sum in the last float bits, so the default keeps the
    pass whole on every rank and the PNG identical to the one-GPU file; SPLIT_PHOTON_MIN_STEPS is the size from which
    the split pays when asked for."""
    if world <= 1 or scene.desc.photon_steps <= 0:
        return False
    return os.environ.get("RT_SPLIT_PHOTONS", "0") not in ("", "0") and scene.desc.photon_steps >= SPLIT_PHOTON_MIN_STEPS


def render_still(renderer, scene):
    """One picture, bands interleaved over the ranks; returns the full [H,W,4] float picture on rank 0."""
    rank, world = world_info()
    h, w = scene.height, scene.width
    bands = capi.Bands(BAND_ROWS, rank, world)
    if split_photon_pass(scene, world):
        # PhotonMapper::generate (src/photonmap.cpp:32-56) once for the whole job: rank r emits the r-th slice
        # of the root photons, the deposits of all slices are summed into every rank's textures
        atlas = renderer.photon_pass(scene, rank, world)
        if atlas is not None:
            settle_for_gloo(atlas)
            dist.all_reduce(atlas, op=dist.ReduceOp.SUM)
        part = renderer.render(scene, bands, photons_ready=True)
    else:
        part = renderer.render(scene, bands)
    # pad to the largest share so that the gather has equal shapes
    n_max = max(len(band_rows_of(h, r, world)) for r in range(world)) * w
    padded = torch.zeros((n_max, 4), dtype=part.dtype, device=part.device)
    padded[: part.shape[0]] = part
    parts = gather_to_root(padded)
    if rank != 0:
        return None
    full = torch.empty((h, w, 4), dtype=part.dtype, device=part.device)
    for r in range(world):
        rows = band_rows_of(h, r, world)
        if rows:
            full[torch.as_tensor(rows, device=part.device)] = parts[r][: len(rows) * w].reshape(len(rows), w, 4)
    return full


def render_units_sequential(renderer, path, units, consume):
    """Lock-step form of render_units (parse, render, blocking gather, consume): kept as the reference the
    pipelined form is tested against (RT_DRIVER_SEQUENTIAL=1 selects it)."""
    rank, world = world_info()
    rounds = math.ceil(len(units) / world)
    for i in range(rounds):
        mine = i * world + rank
        if mine < len(units):
            pic = renderer.render(capi.Scene(path, units[mine][2]))
        else:
            probe = capi.Scene(path, units[0][2])      # a rank without a unit in the last round still takes part
            pic = torch.zeros((probe.width * probe.height, 4), dtype=torch.float32, device=getattr(renderer, "device", torch.device("cpu")))
        pics = gather_to_root(pic)
        if rank == 0:
            for r in range(world):
                j = i * world + r
                if j < len(units):
                    consume(units[j], pics[r])


def render_units(renderer, path, units, consume, payload=None):
    """Renders `units` [(key..., t, ...)] round-robin over the ranks; on rank 0 calls consume(unit, picture) in unit
    order.  Pipelined: the scene file of the next unit is parsed on a host thread while this one renders, the gather
    of round i (async, two buffers) runs while round i+1 renders, and rank 0 consumes round i-1 meanwhile.
    `payload(picture)` is what travels (e.g. the picture quantised to RGBA8 on the rank that rendered it: a quarter
    of the bytes); default the float picture."""
    if os.environ.get("RT_DRIVER_SEQUENTIAL", "0") not in ("", "0"):
        if payload is None:
            return render_units_sequential(renderer, path, units, consume)
        return render_units_sequential(renderer, path, units, lambda u, pic: consume(u, payload(pic)))
    from concurrent.futures import ThreadPoolExecutor
    rank, world = world_info()
    rounds = math.ceil(len(units) / world)
    device = getattr(renderer, "device", torch.device("cpu"))
    payload = payload or (lambda pic: pic)
    bufs, gathered, pending = [None, None], [None, None], [None, None]
    # RT_DRIVER_FORCE_GATHER=1 sends a one-rank group through the collective too (hardware test of the RCCL path)
    collective = world > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get("RT_DRIVER_FORCE_GATHER", "0") not in ("", "0"))

    def load(i):
        mine = i * world + rank
        return capi.Scene(path, units[mine][2]) if mine < len(units) else None

    def consume_round(i):
        if pending[i % 2] is not None:
            pending[i % 2].wait()
            pending[i % 2] = None
        if rank == 0:
            pics = gathered[i % 2] if collective else [bufs[i % 2]]
            for r in range(world):
                j = i * world + r
                if j < len(units):
                    consume(units[j], pics[r])

    lanes = renderer.lanes() if hasattr(renderer, "lanes") else [(renderer, None)]
    n_lanes = len(lanes)
    raw = [None] * n_lanes    # several lanes: (picture, scene) of the units whose photon pass has not been confirmed yet
    unconfirmed = [0]         # the oldest of them

    def finish(i, out):
        """Round i's payload is on its way: hand it to the gather, after consuming round i-2 (whose buffers it takes)."""
        if i >= 2:
            consume_round(i - 2)
        b = i % 2
        bufs[b] = out
        if collective:
            if rank == 0:
                # fresh tensors every round (the caching allocator makes that cheap): consumers may still be reading
                # the previous ones on another stream (HostFrames), which record_stream keeps alive, not unwritten
                gathered[b] = [torch.empty_like(out) for _ in range(world)]
            settle_for_gloo(out)
            pending[b] = dist.gather(out, gathered[b] if rank == 0 else None, dst=0, async_op=True)

    def confirm_up_to(last):
        """Several lanes: a unit is enqueued with its photon pass unchecked (asking waits for the pass, and the host has the
        next units to upload meanwhile).  Confirms the units up to `last` in order, renders one again in the rare case
        that its pass outgrew the record buffer, and sends the payloads on."""
        while unconfirmed[0] <= last:
            i = unconfirmed[0]
            unconfirmed[0] += 1
            lane, side = lanes[i % n_lanes]
            pic, scene = raw[i % n_lanes]
            raw[i % n_lanes] = None
            main = torch.cuda.current_stream()
            with torch.cuda.stream(side):
                if not lane.photon_pass_ok():
                    pic = lane.render(scene)
                out = payload(pic)
            out.record_stream(main)
            main.wait_stream(side)          # what follows on the main stream (gather, consume) sees the finished picture
            finish(i, out)

    with ThreadPoolExecutor(max_workers=1) as pool:
        nxt = pool.submit(load, 0) if rounds else None
        for i in range(rounds):
            scene = nxt.result()
            nxt = pool.submit(load, i + 1) if i + 1 < rounds else None
            if scene is None:
                # a rank without a unit in the last round still takes part in the gather
                if n_lanes > 1:
                    confirm_up_to(i - 1)
                    unconfirmed[0] = i + 1
                probe = capi.Scene(path, units[0][2])
                finish(i, payload(torch.zeros((probe.width * probe.height, 4), dtype=torch.float32, device=device)))
            elif n_lanes > 1:
                lane, side = lanes[i % n_lanes]
                with torch.cuda.stream(side):
                    raw[i % n_lanes] = (lane.render(scene, defer_check=True), scene)
                confirm_up_to(i - (n_lanes - 1))      # the lane of unit i+1 becomes free; rounds up to i are on the GPU's queue
            else:
                finish(i, payload(renderer.render(scene)))
        if n_lanes > 1:
            confirm_up_to(rounds - 1)
        for i in range(max(0, rounds - 2), rounds):
            consume_round(i)


class HostFrames:
    """RGBA8 frames on their way to the host: on a GPU each is copied into pinned memory on a side stream, so the
    rank that writes the file never stalls its render queue on a download; frames() waits for them all."""

    def __init__(self, device):
        self.items = []
        self.stream = torch.cuda.Stream(device) if device.type == "cuda" else None

    def add(self, q):
        if self.stream is None:
            self.items.append(q.cpu())
            return
        host = torch.empty(q.shape, dtype=q.dtype, pin_memory=True)
        self.stream.wait_event(torch.cuda.current_stream().record_event())
        with torch.cuda.stream(self.stream):
            host.copy_(q, non_blocking=True)
        q.record_stream(self.stream)
        self.items.append(host)

    def frames(self):
        if self.stream is not None:
            self.stream.synchronize()
        return [t.numpy() for t in self.items]


def render_frames(renderer, path):
    """The loop part of the reference's main(): picks one of the four loops from the scene's settings and returns
    (RGBA8 frames on rank 0 / None elsewhere, scene info, per-frame fps or None for a still)."""
    rank, world = world_info()
    probe = capi.Scene(path, 0.0)
    info = probe.info
    cam = probe.desc.camera
    if cam.supersampling <= 1 and cam.lens_size != 0.0:
        raise RuntimeError("Depth of field needs supersampling.")
    h, w = probe.height, probe.width
    animated = info.frames > 1 and info.still_time == f32(capi.FLT_MAX)
    host = HostFrames(getattr(renderer, "device", torch.device("cpu")))
    fps = None

    def to_host8(rgba):
        host.add(renderer.quantize(rgba.reshape(h, w, 4).contiguous()))

    if not animated and info.sub_frames <= 1:
        scene = capi.Scene(path, float(still_time(info)))
        full = render_still(renderer, scene)
        if rank == 0:
            to_host8(full)
    elif info.sub_frames <= 1:
        units = [(f, 0, float(f32(f) / f32(info.frames - 1))) for f in range(info.frames)]
        # every rank quantises its own frame (Color::scaleOut): RGBA8 travels, a quarter of the float picture
        render_units(renderer, path, units, lambda u, q: host.add(q.reshape(h, w, 4)),
                     payload=lambda pic: renderer.quantize(pic.reshape(h, w, 4).contiguous()))
        if rank == 0:
            # writeAPNGFrame(..., scene.fps()) with the frame's own scene (src/main.cpp:105): fps may be keyframed
            fps = [capi.Scene(path, u[2]).info.fps for u in units]
    else:
        units = motion_blur_units(path, info)
        state = {"acc": None}

        def consume(u, pic):
            _frame, sub, _t, weight, k = u
            if state["acc"] is None:
                state["acc"] = torch.zeros_like(pic)
            renderer.mul_add(state["acc"], pic, weight)      # Picture::mulAdd, in sub-frame order
            if sub == k - 1:
                to_host8(state["acc"])
                state["acc"] = None

        render_units(renderer, path, units, consume)
        if animated:
            fps = [info.fps] * (units[-1][0] + 1)      # renderVideoMotionBlur uses origScene.fps() (src/main.cpp:152)
    frames8 = host.frames()
    return (frames8 if rank == 0 else None), info, fps


def render_scene_file(renderer, path, out_path=None):
    """≙ main() of the reference: renders the file's still or animation and writes the PNG/APNG on rank 0.
    Returns the list of RGBA8 frames on rank 0 (None elsewhere)."""
    frames8, info, fps = render_frames(renderer, path)
    if frames8 is None:
        return None
    out_path = out_path or info.output_file.decode()
    if fps is not None:
        capi.apng_write(out_path, frames8, fps)
    else:
        capi.png_write(out_path, frames8[0])
    return frames8


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="rt_amd-compatible renderer on 1..N GPUs (launch with torchrun for N > 1)")
    ap.add_argument("scene")
    ap.add_argument("out", nargs="?")
    args = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RT_DIST_BACKEND", "nccl") == "gloo":
        local_rank %= max(1, torch.cuda.device_count())      # rehearsal: more ranks than GPUs, collectives through gloo
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("RT_DIST_BACKEND", "nccl") == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    try:
        renderer = HipRenderer(local_rank)
        if world > 1:
            # the gather of one unit runs beside the render of the next: its kernels sit on the GPU when that grid is
            # dispatched (include/rt_hip.h, RT_OPT_SHARED_GPU; frames as small as C5's are not affected either way)
            renderer.ctx.set_option(capi.RT_OPT_SHARED_GPU, 1)
        render_scene_file(renderer, args.scene, args.out)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
