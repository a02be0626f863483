"""The scene pipeline of the eval path: what `bench.py` measures and what the eval runners run (one implementation).

Reference loop it replaces: llava/eval/model_scanqa.py:130-206 - per question: load the scene's frames on the host, run the whole
model, decode token by token, all on one thread and one stream.  Here, per GPU:

  * `AsyncSceneLoader`: worker processes (v3d.frame_io) decode the JPEG / depth-PNG / pose-txt files of the NEXT questions' scenes
    straight into pinned shared-memory blocks while the GPU works (video_utils.py:196-238, 285-290 are host I/O).  A scene asked
    about again shortly after is not decoded twice.
  * `ScenePipeline.prefill`: upload (pinned, asynchronous) -> back-projection at the surviving pixels (K1+K2), Pillow-exact RGB
    resize + crop + normalise (a6/a7) -> ViT -> projector -> fusion -> Qwen2 prefill, on stream A, one scene after the other
    (MFMA-bound).  The device tensors of the last few scenes are kept, so consecutive questions about one scene upload once.
  * `ScenePipeline.run`: scenes decode in groups of up to 16 on stream B (`Engine.decode_group`: one pass over the weights per
    step for the whole group, HBM-bound) while stream A already prefills the next group into the other context set; the stop
    test runs on the device and the host polls it two steps late, never per token.

Every (scene, question) still takes the complete path; a scene's tokens do not depend on its group (tests/test_gpu_engine.py).
"""
import collections
import time
from dataclasses import dataclass, field
from typing import Any, Optional

import torch

from . import ops
from ._native import V3DError


@dataclass
class SceneSample:
    """One (scene, question).  Either processed tensors (`images` [F,3,S,S] + `world_coords` [F,S,S,3], what
    VideoProcessor.process_3d_video returns) or raw host arrays (`raw`: AsyncSceneLoader's payload)."""
    input_ids: torch.Tensor                     # 1-D, host, exactly one IMAGE_TOKEN_INDEX
    images: Optional[torch.Tensor] = None
    world_coords: Optional[torch.Tensor] = None
    raw: Any = None                             # dict(depth int16-view [F,H,W], K [F,4,4], pose [F,4,4], frames u8 [F,H,W,3]) or a future of it
    key: Any = None                             # scene id: device tensors of equal keys are shared
    box_input: Optional[torch.Tensor] = None
    extra: dict = field(default_factory=dict)


class _HostBlock:
    """One scene's frames in a shared-memory block that worker processes write and the GPU reads by DMA (registered as pinned)."""

    def __init__(self, nbytes):
        from multiprocessing import shared_memory
        self.shm = shared_memory.SharedMemory(create=True, size=nbytes)
        self.nbytes = self.shm.size
        self.bytes = torch.frombuffer(self.shm.buf, dtype=torch.uint8)
        self.pinned = False
        if torch.cuda.is_available():
            rc = torch.cuda.cudart().cudaHostRegister(self.bytes.data_ptr(), self.nbytes, 0)
            self.pinned = int(rc) == 0
        self.event = None                   # the last upload out of this block

    def tensors(self, layout):
        out = {}
        for k, (shape, dt, off) in layout.items():
            n = 1
            for d in shape:
                n *= d
            tdt = getattr(torch, dt)
            out[k] = self.bytes[off: off + n * torch.empty(0, dtype=tdt).element_size()].view(tdt).view(shape)
        return out

    def close(self):
        try:
            if self.pinned:
                torch.cuda.cudart().cudaHostUnregister(self.bytes.data_ptr())
        finally:
            self.bytes = None
            try:
                self.shm.close()
            except BufferError:
                pass
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass


class AsyncSceneLoader:
    """Decodes the frames of the scenes of `keys` (one key per question, in question order) `ahead` scenes in front of the consumer,
    in worker PROCESSES (v3d.frame_io) that write straight into pinned shared-memory blocks.
    describe(key) -> dict(files=[colour file of each frame], axis_align=4x4, K=[4,4] f32 tensor[, clamp=(lo, hi)]) - called on the
    consumer's thread.
    get(j) -> (payload, seconds waited): payload = dict(depth int16-view [F,Hd,Wd], frames u8 [F,Hc,Wc,3], pose f32 [F,4,4], K [F,4,4],
    _done=callback(event)) - the consumer calls payload["_done"](event recorded after its uploads, or None) when it no longer needs
    the host arrays; the block is then recycled.  Equal keys within `keep` scenes of each other are decoded once.
    workers = 0: no processes, frames are decoded on the consumer's thread inside get() (tests, tiny runs)."""

    def __init__(self, keys, describe, workers=8, ahead=6, keep=4, pool=None):
        from . import frame_io
        self.io = frame_io
        self.keys, self.describe = list(keys), describe
        if pool is None and workers > 0 and torch.cuda.is_available() and torch.cuda.is_initialized():
            # forking a process whose HIP runtime threads and pinned mappings are live is fragile (frame_io.make_pool: make the pool
            # BEFORE the GPU is touched and pass it in); without one the frames are decoded on the calling thread
            import warnings
            warnings.warn("AsyncSceneLoader: no decoding pool was passed and the GPU is already initialised - decoding on the calling "
                          "thread (make the pool with v3d.frame_io.make_pool before the first GPU call and pass pool=...)", stacklevel=2)
            workers = 0
        self.own_pool = pool is None and workers > 0
        self.pool = pool if pool is not None else (frame_io.make_pool(workers) if workers > 0 else None)
        self.ahead, self.keep = ahead, keep
        self.jobs = {}
        self.by_key = collections.OrderedDict()
        self.free, self.blocks = [], []
        self.next_submit = 0
        self.stage_seconds = collections.Counter()

    def _block(self, nbytes):
        for i, blk in enumerate(self.free):
            if blk.nbytes >= nbytes:
                self.free.pop(i)
                if blk.event is not None:
                    blk.event.synchronize()         # its last upload has left the block
                    blk.event = None
                return blk
        blk = _HostBlock(nbytes)
        self.blocks.append(blk)
        return blk

    def _unref(self, job, event=None):
        if event is not None:
            job["block"].event = event
        job["refs"] -= 1
        if job["refs"] == 0:
            self.free.append(job["block"])

    def _submit(self, j):
        key = self.keys[j]
        job = self.by_key.get(key)
        if job is not None:
            self.by_key.move_to_end(key)
            job["refs"] += 1
            self.jobs[j] = job
            return
        d = self.describe(key)
        files = list(d["files"])
        color_hw, depth_hw = self.io.image_sizes(files[0])
        layout, nbytes = self.io.scene_layout(len(files), color_hw, depth_hw)
        blk = self._block(nbytes)
        payload = blk.tensors(layout)
        payload["K"] = d["K"].float().unsqueeze(0).repeat(len(files), 1, 1)
        payload["clamp"] = d.get("clamp")           # ([lo x, y, z], [hi x, y, z]) under a 'norm' frame-sampling strategy, else None
        align = [list(map(float, r)) for r in d["axis_align"]]
        job = {"payload": payload, "block": blk, "refs": 2, "layout": layout, "files": files, "align": align, "futures": None}     # this position + the key table
        if self.pool is not None:
            job["futures"] = [self.pool.submit(self.io._worker_decode, blk.shm.name, layout, i, f, align) for i, f in enumerate(files)]
        self.jobs[j] = job
        self.by_key[key] = job
        while len(self.by_key) > self.keep:
            _, old = self.by_key.popitem(last=False)
            self._unref(old)

    def get(self, j):
        while self.next_submit < min(len(self.keys), j + 1 + self.ahead):
            self._submit(self.next_submit)
            self.next_submit += 1
        job = self.jobs.pop(j)
        t0 = time.perf_counter()
        if job["futures"] is not None:
            for f in job["futures"]:
                self.stage_seconds.update(f.result())         # re-raises a worker's exception here, on the consumer's thread
            job["futures"] = []
        elif job["files"] is not None:                         # inline mode
            arrays = self.io.views(job["block"].shm.buf, job["layout"])
            for i, f in enumerate(job["files"]):
                self.stage_seconds.update(self.io.decode_into(arrays, i, f, job["align"]))
            job["files"] = None
        payload = dict(job["payload"])
        payload["_done"] = lambda event=None, job=job: self._unref(job, event)
        return payload, time.perf_counter() - t0

    def close(self):
        if self.own_pool and self.pool is not None:
            self.pool.shutdown(wait=True, cancel_futures=True)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        for blk in self.blocks:
            blk.close()
        self.blocks, self.free = [], []


class ScenePipeline:
    def __init__(self, eng, group_size=16, crop=384, scene_cache=3, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5),
                 rescale=1 / 255, prefill_streams=1, decode_contexts=True):
        if not 1 <= group_size <= eng.MAX_GROUP:
            raise V3DError(f"decode groups hold 1 to {eng.MAX_GROUP} scenes")
        self.eng, self.G, self.crop = eng, group_size, crop
        self.mean, self.std, self.rescale = tuple(image_mean), tuple(image_std), rescale
        # decode_contexts=False (SceneReusePipeline): only the device-input half is wanted - no per-scene decode contexts (0.6 GB each)
        self.sets = [[eng.ctx] + [eng.new_context() for _ in range(group_size - 1)], [eng.new_context() for _ in range(group_size)]] \
            if decode_contexts else None
        self.groups = [eng.new_group(group_size), eng.new_group(group_size)] if decode_contexts else None
        dev = torch.device(eng.device)
        self.streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        self.copy_stream = torch.cuda.Stream(device=dev)
        # prefill_streams = 2: consecutive scenes prefill on two streams with their own scratch, so that one scene's kernels run on the
        # CUs the other's last GEMM rounds leave idle (a persistent 256-workgroup GEMM ends with a partly filled round)
        self.pre_streams = [self.streams[0]] + [torch.cuda.Stream(device=dev) for _ in range(prefill_streams - 1)]
        self.workspaces = [eng.ws] + [eng.new_prefill_workspace() for _ in range(prefill_streams - 1)]
        self.scene_cache = collections.OrderedDict()
        self.scene_cache_size = scene_cache
        self.upload_seconds = 0.0
        self.wait_seconds = 0.0

    # ------------------------------------------------------------------ one scene's question-independent device inputs
    def device_inputs(self, sample):
        """-> (pixel_values [F,3,crop,crop], world_coords [F,crop,crop,3]) in the engine's dtype, on the current stream."""
        eng, dt = self.eng, self.eng.dtype
        if sample.key is not None and sample.key in self.scene_cache:
            self.scene_cache.move_to_end(sample.key)
            if isinstance(sample.raw, dict) and "_done" in sample.raw:
                sample.raw["_done"](None)          # nothing to upload: the scene's device tensors are still here
            images, coords, ready = self.scene_cache[sample.key]
            torch.cuda.current_stream().wait_event(ready)       # (they may have been produced on the other prefill stream)
            return images, coords
        if sample.images is not None:
            images = sample.images.to(device=eng.device, dtype=dt, non_blocking=True)
            coords = sample.world_coords.to(device=eng.device, dtype=dt, non_blocking=True)
        else:
            raw = sample.raw
            if hasattr(raw, "result"):
                raw = raw.result()
            t0 = time.perf_counter()
            # uploads ride their own stream: the 140 MB of the NEXT scene cross PCIe while this stream still runs the previous scene's
            # kernels (the host is a scene or more ahead of the GPU)
            cur = torch.cuda.current_stream()
            with torch.cuda.stream(self.copy_stream):
                depth = raw["depth"].to(eng.device, non_blocking=True)
                K = raw["K"].to(eng.device, non_blocking=True)
                pose = raw["pose"].to(eng.device, non_blocking=True)
                frames = raw["frames"].to(eng.device, non_blocking=True)
                up = self.copy_stream.record_event()
            for t in (depth, K, pose, frames):
                t.record_stream(cur)
            cur.wait_event(up)
            if "_done" in raw:                     # the loader may recycle the host block once these copies have run
                raw["_done"](up)
            self.upload_seconds += time.perf_counter() - t0
            coords = ops.unproject_sampled(depth, K, pose, self.crop, dt)                                     # K1 + K2
            bounds = raw.get("clamp")                            # 'norm' strategies: calculate_world_coords(do_normalize=True), :232-236
            if bounds is not None:                               # (clamping commutes with the nearest-neighbour gather)
                ops.clamp_xyz(coords, bounds[0], bounds[1])
            if tuple(frames.shape[1:3]) == (self.crop, self.crop):
                images = ops.preprocess_rgb(frames, dt, self.mean, self.std, self.rescale)
            else:                                                # video_utils.py:297-306 + siglip_encoder.py:47-67 in one kernel
                # the DEPTH map's size steers the colour resize too: V, H, W come from world_coords (video_utils.py:264, 298-299), so
                # ScanNet's 1296 x 968 colour frames go to 512 x 384 (640 x 480 depth), not to the 514 their own aspect would give
                Hd, Wd = depth.shape[1:3]
                new_w = int(Wd * (self.crop / Hd))
                images = ops.resize_crop_rgb(frames, (self.crop, new_w), crop=(0, (new_w - self.crop) // 2, self.crop, self.crop),
                                             dtype=dt, mean=self.mean, std=self.std, rescale=self.rescale)
        if sample.key is not None and self.scene_cache_size > 0:
            for t in (images, coords):
                for st in self.pre_streams:
                    t.record_stream(st)
            self.scene_cache[sample.key] = (images, coords, torch.cuda.current_stream().record_event())
            while len(self.scene_cache) > self.scene_cache_size:
                self.scene_cache.popitem(last=False)
        return images, coords

    def prefill(self, sample, max_new_tokens, stamps=None):
        """geometry -> ViT -> projector -> fusion -> Qwen2 prefill of one (scene, question) into the engine's current context;
        returns the prompt length S.  The last row's logits stay in ctx.logits[0]."""
        eng = self.eng
        images, coords = self.device_inputs(sample)
        feats = eng.encode_images(images)
        ids = eng.voxel_ids(coords)
        pe = (stamps["pe"] if stamps else None)
        x = eng.build_inputs_embeds(sample.input_ids, feats, ids, box_input=sample.box_input,
                                    coord_token_id=sample.extra.get("coord_token_id"), stamp=pe)
        S = x.shape[0]
        eng._check_room(S, max_new_tokens)
        eng.llm_forward(x, 0, stamps=stamps, last_rows=[S - 1])
        return S

    # ------------------------------------------------------------------ the whole list
    @torch.no_grad()
    def run(self, samples, max_new_tokens, eos_token_id=None, overlap=True, stamps=None, trim=True):
        """samples: iterable of SceneSample (consumed lazily, one decode group ahead).  Returns one host LongTensor of new token ids
        per sample, in order, cut after its first EOS (trim=False: the untrimmed [n, steps] rows per group, for benchmarking)."""
        eng, G = self.eng, self.G
        sA, sB = self.streams
        it = iter(samples)
        keep = eng.ctx
        out = []

        def prefill_group(gi, first):
            """Pulls up to G samples from the iterator, one at a time (a sample's host block is uploaded before the next one is
            asked for), and queues their prefills on stream A.  None when the iterator is exhausted."""
            ctxs_all = self.sets[gi % 2]
            ctxs, lens, used, alone = [], [], set(), None
            n_pre = len(self.pre_streams) if overlap else 1
            for k, c in enumerate(ctxs_all):
                smp = next(it, None)
                if smp is None:
                    break
                st = self.pre_streams[k % n_pre] if overlap else torch.cuda.current_stream()
                with torch.cuda.stream(st):
                    eng.use_workspace(self.workspaces[k % n_pre])
                    eng.use(c)
                    stamped = stamps is not None and first and not lens
                    if stamps is not None and first and len(lens) == 1:
                        st.wait_event(alone)        # the stamped scene ran with nothing else on the chip
                    # kernel stamps on the first scene only: its prefill runs with nothing else on the chip
                    lens.append(self.prefill(smp, max_new_tokens, stamps if stamped else None))
                    if stamped:
                        alone = st.record_event()
                ctxs.append(c)
                used.add(st)
            done = [st.record_event() for st in used]
            eng.use_workspace(self.workspaces[0])
            return (ctxs, lens, done) if ctxs else None

        try:
            cur = torch.cuda.current_stream()
            if overlap:
                for st in self.pre_streams:
                    st.wait_stream(cur)
                sB.wait_stream(cur)
            gi = 0
            pending = prefill_group(0, True)
            while pending is not None:
                ctxs, lens, pre_done = pending
                # the next group's prefills are queued BEFORE this group's decode: the host then polls the decode's stop test
                # while stream A stays busy.  Its context set was last used by group gi - 1, whose tokens the host already holds.
                pending = prefill_group(gi + 1, False)
                with torch.cuda.stream(sB if overlap else torch.cuda.current_stream()):
                    for ev in pre_done:
                        torch.cuda.current_stream().wait_event(ev)
                    toks = eng.decode_group(self.groups[gi % 2], ctxs, lens, max_new_tokens, eos_token_id=eos_token_id)
                    out += eng.trim_at_eos(toks, eos_token_id) if trim else [toks.cpu()]
                gi += 1
            if overlap:
                for st in self.pre_streams:
                    cur.wait_stream(st)
                cur.wait_stream(sB)
        finally:
            eng.use_workspace(self.workspaces[0])      # (a prefill that raised part-way may have left the second one selected)
            eng.use(keep)
        return out


class SceneReusePipeline:
    """Scene-level reuse AND the pipeline (r04; VERDICT r03 missing #2): consecutive questions about one scene share ONE
    scene prefill (Engine.prefill_scene: geometry, ViT, projector, fusion, the decoder over [system | user | <image>]) and are answered in
    batches (Engine.answer_group: the questions' rows as one batch over the cached prefix, then one decode group) - with the asynchronous
    host loader in front and scene i+1's upload / device inputs / ViT / prefix prefill (MFMA-bound, stream A, its own scratch and scene
    context) queued BEFORE scene i's answer groups (weight streaming, HBM-bound, stream B, the other scratch), so the two run side by side.
    The reference recomputes the whole prompt for every question, one after the other (model_scanqa.py:130-206).
    Records are those of the synchronous form (eval_scanqa.scene_batches): same kernels on the same operands; the scratch and the stream
    a launch goes to do not enter the arithmetic (tests/test_gpu_eval_harness.py)."""

    def __init__(self, eng, crop=384, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), rescale=1 / 255, batch=32):
        if not 1 <= batch <= eng.MAX_GROUP:
            raise V3DError(f"answer batches hold 1 to {eng.MAX_GROUP} questions")
        self.eng, self.batch = eng, batch
        self.inputs = ScenePipeline(eng, 1, crop=crop, scene_cache=0, image_mean=image_mean, image_std=image_std, rescale=rescale,
                                    decode_contexts=False)
        dev = torch.device(eng.device)
        self.sA, self.sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        self.scene_ctx = [eng.ctx, eng.new_context()]            # the prefix K/V of the scene being answered / being prefilled
        self.ws = [eng.ws, eng.new_prefill_workspace()]          # [0]: scene prefills (stream A); [1]: the question-row batches (stream B)
        self.wait_seconds = 0.0

    @property
    def upload_seconds(self):
        return self.inputs.upload_seconds

    @torch.no_grad()
    def run(self, scenes, max_new_tokens, eos_token_id=None, overlap=True):
        """scenes: iterable of (SceneSample whose input_ids are the PREFIX ids up to and including <image>, [question id tensors]) - consumed
        lazily, one scene ahead.  Returns, per scene, the list of its answers' token-id tensors (host, cut after the first EOS)."""
        eng = self.eng
        keep_ctx, keep_ws = eng.ctx, eng.ws
        it = iter(scenes)
        cur = torch.cuda.current_stream()
        sA, sB = (self.sA, self.sB) if overlap else (cur, cur)
        out = []

        def queue_prefill(k):
            item = next(it, None)
            if item is None:
                return None
            smp, questions = item
            with torch.cuda.stream(sA):
                eng.use_workspace(self.ws[0])
                eng.use(self.scene_ctx[k % 2])
                images, coords = self.inputs.device_inputs(smp)
                P = eng.prefill_scene(smp.input_ids, images, coords)
                done = sA.record_event()
            return k, questions, P, done

        try:
            if overlap:
                sA.wait_stream(cur)
                sB.wait_stream(cur)
            pending = queue_prefill(0)
            while pending is not None:
                k, questions, P, done = pending
                # the NEXT scene's prefill goes to stream A before this scene's answers are queued: the host then sits in the answer
                # groups' stop-test polling while stream A works.  Its context was last read by scene k - 1's answers, which the host
                # has already collected.
                pending = queue_prefill(k + 1)
                answers = []
                with torch.cuda.stream(sB):
                    sB.wait_event(done)
                    eng.use_workspace(self.ws[1])
                    eng.use(self.scene_ctx[k % 2])
                    for b0 in range(0, len(questions), self.batch):
                        qs = questions[b0: b0 + self.batch]
                        room = eng.cfg.llm.max_pos - P - max(int(q.numel()) for q in qs) + 1
                        answers += [a.cpu() for a in eng.answer_group(qs, max_new_tokens=min(max_new_tokens, room), eos_token_id=eos_token_id)]
                out.append(answers)
            if overlap:
                cur.wait_stream(sA)
                cur.wait_stream(sB)
        finally:
            eng.use_workspace(keep_ws)
            eng.use(keep_ctx)
        return out
