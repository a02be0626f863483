"""The scene pipeline of the eval path: what `bench.py` measures and what the eval runners run (one implementation).

Reference loop it replaces: llava/eval/model_scanqa.py:130-206 - per question: load the scene's frames on the host, run the whole
model, decode token by token, all on one thread and one stream.  Here, per GPU:

  * `AsyncSceneLoader`: a thread pool decodes the JPEG / depth-PNG / pose-txt files of the NEXT questions' scenes into pinned
    buffers while the GPU works (video_utils.py:196-238, 285-290 are host I/O; PIL releases the GIL while it decodes).  A scene
    asked about again shortly after is not decoded twice.
  * `ScenePipeline.prefill`: upload (pinned, asynchronous) -> back-projection at the surviving pixels (K1+K2), Pillow-exact RGB
    resize + crop + normalise (a6/a7) -> ViT -> projector -> fusion -> Qwen2 prefill, on stream A, one scene after the other
    (MFMA-bound).  The device tensors of the last few scenes are kept, so consecutive questions about one scene upload once.
  * `ScenePipeline.run`: scenes decode in groups of up to 16 on stream B (`Engine.decode_group`: one pass over the weights per
    step for the whole group, HBM-bound) while stream A already prefills the next group into the other context set; the stop
    test runs on the device and the host polls it two steps late, never per token.

Every (scene, question) still takes the complete path; a scene's tokens do not depend on its group (tests/test_gpu_engine.py).
"""
import collections
import concurrent.futures as cf
import threading
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


class AsyncSceneLoader:
    """Loads `load_frame(key, i, out)` for the frames of each key on a thread pool, `ahead` scenes in front of the consumer.
    plan(key) -> (n_frames, alloc) where alloc() returns the dict of pinned buffers load_frame fills; get(j) blocks until scene j
    (in the order of `keys`) is complete and returns (payload, seconds the consumer waited).  Equal keys within `keep` scenes of each
    other are loaded once."""

    def __init__(self, keys, plan, load_frame, workers=8, ahead=6, keep=4):
        self.keys, self.plan, self.load_frame = list(keys), plan, load_frame
        self.pool = cf.ThreadPoolExecutor(max_workers=workers, thread_name_prefix="v3d-loader")
        self.ahead, self.keep = ahead, keep
        self.jobs = {}                      # position -> (payload, [futures])
        self.by_key = collections.OrderedDict()
        self.next_submit = 0
        self.stage_seconds = collections.Counter()
        self.lock = threading.Lock()

    def _submit(self, j):
        key = self.keys[j]
        hit = self.by_key.get(key)
        if hit is not None:
            self.by_key.move_to_end(key)
            self.jobs[j] = hit
            return
        n, alloc = self.plan(key)
        payload = alloc()

        def one(i):
            t = self.load_frame(key, i, payload)
            if t:
                with self.lock:
                    self.stage_seconds.update(t)

        job = (payload, [self.pool.submit(one, i) for i in range(n)])
        self.jobs[j] = job
        self.by_key[key] = job
        while len(self.by_key) > self.keep:
            self.by_key.popitem(last=False)

    def get(self, j):
        while self.next_submit < min(len(self.keys), j + 1 + self.ahead):
            self._submit(self.next_submit)
            self.next_submit += 1
        payload, futs = self.jobs.pop(j)
        t0 = time.perf_counter()
        for f in futs:
            f.result()                      # re-raises a loader exception here, on the consumer's thread
        return payload, time.perf_counter() - t0

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)


class ScenePipeline:
    def __init__(self, eng, group_size=16, crop=384, scene_cache=3, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5),
                 rescale=1 / 255):
        if not 1 <= group_size <= 16:
            raise V3DError("decode groups hold 1 to 16 scenes")
        self.eng, self.G, self.crop = eng, group_size, crop
        self.mean, self.std, self.rescale = tuple(image_mean), tuple(image_std), rescale
        self.sets = [[eng.ctx] + [eng.new_context() for _ in range(group_size - 1)], [eng.new_context() for _ in range(group_size)]]
        self.groups = [eng.new_group(group_size), eng.new_group(group_size)]
        dev = torch.device(eng.device)
        self.streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
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
            return self.scene_cache[sample.key]
        if sample.images is not None:
            images = sample.images.to(device=eng.device, dtype=dt, non_blocking=True)
            coords = sample.world_coords.to(device=eng.device, dtype=dt, non_blocking=True)
        else:
            raw = sample.raw
            if hasattr(raw, "result"):
                raw = raw.result()
            t0 = time.perf_counter()
            depth = raw["depth"].to(eng.device, non_blocking=True)
            K = raw["K"].to(eng.device, non_blocking=True)
            pose = raw["pose"].to(eng.device, non_blocking=True)
            frames = raw["frames"].to(eng.device, non_blocking=True)
            self.upload_seconds += time.perf_counter() - t0
            coords = ops.unproject_sampled(depth, K, pose, self.crop, dt)                                     # K1 + K2
            H, W = frames.shape[1:3]
            if (H, W) == (self.crop, self.crop):
                images = ops.preprocess_rgb(frames, dt, self.mean, self.std, self.rescale)
            else:                                                # video_utils.py:297-306 + siglip_encoder.py:47-67 in one kernel
                new_w = int(W * (self.crop / H))
                images = ops.resize_crop_rgb(frames, (self.crop, new_w), crop=(0, (new_w - self.crop) // 2, self.crop, self.crop),
                                             dtype=dt, mean=self.mean, std=self.std, rescale=self.rescale)
        if sample.key is not None and self.scene_cache_size > 0:
            self.scene_cache[sample.key] = (images, coords)
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
        eng.llm_forward(x, 0, stamps=stamps)
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

        def take():
            batch = []
            for smp in it:
                batch.append(smp)
                if len(batch) == G:
                    break
            return batch

        def prefill_group(gi, batch, first):
            ctxs = self.sets[gi % 2][: len(batch)]
            lens = []
            with torch.cuda.stream(sA if overlap else torch.cuda.current_stream()):
                for c, smp in zip(ctxs, batch):
                    eng.use(c)
                    # kernel stamps on the first scene only: its prefill runs with nothing else on the chip
                    lens.append(self.prefill(smp, max_new_tokens, stamps if (first and not lens) else None))
                done = torch.cuda.current_stream().record_event()
            return ctxs, lens, done

        try:
            cur = torch.cuda.current_stream()
            if overlap:
                sA.wait_stream(cur)
                sB.wait_stream(cur)
            batch = take()
            gi = 0
            pending = prefill_group(0, batch, True) if batch else None
            while pending is not None:
                ctxs, lens, pre_done = pending
                nxt = take()
                # the next group's prefills are queued BEFORE this group's decode: the host then polls the decode's stop test
                # while stream A stays busy.  Its context set was last used by group gi - 1, whose tokens the host already holds.
                pending = prefill_group(gi + 1, nxt, False) if nxt else None
                with torch.cuda.stream(sB if overlap else torch.cuda.current_stream()):
                    torch.cuda.current_stream().wait_event(pre_done)
                    toks = eng.decode_group(self.groups[gi % 2], ctxs, lens, max_new_tokens, eos_token_id=eos_token_id)
                    out += eng.trim_at_eos(toks, eos_token_id) if trim else [toks.cpu()]
                gi += 1
            if overlap:
                cur.wait_stream(sA)
                cur.wait_stream(sB)
        finally:
            eng.use(keep)
        return out
