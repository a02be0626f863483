"""Host I/O of one frame of a scene - the file reads of llava/video_utils.py:214-227 (depth PNG, pose txt) and :285-290 (RGB file) -
as plain functions over numpy arrays, importable WITHOUT torch so that worker processes can run them.

Why processes: decoding the 32 x (1296 x 968 JPEG + 640 x 480 16-bit PNG) of one question costs ~0.5 s of one core; with loader
THREADS the conversions that hold the GIL (PIL -> numpy) and the GPU-launching main thread starve each other (measured: 16 threads
spent 1.7 s of thread time per question, a pose file's 20 us parse waited 2.4 ms for the GIL, and the launch rate of the main thread
halved the GPU's throughput).  Worker processes decode straight into a shared-memory block the parent has registered as pinned
host memory, so the parent does no per-pixel work at all: it enqueues one asynchronous copy per array.
"""
import time
from multiprocessing import shared_memory

import numpy as np
from PIL import Image


def read_depth(path_jpg):
    """video_utils.py:214-217: the 16-bit depth PNG beside the colour file -> uint16 [H, W] (millimetres)."""
    with Image.open(path_jpg.replace(".jpg", ".png")) as im:
        return np.array(im).astype(np.uint16)


def read_matrix(path, shape=(4, 4)):
    """np.loadtxt(path) value for value (each field through a correctly rounded decimal -> double conversion), without loadtxt's
    Python-level overhead."""
    with open(path) as f:
        return np.array([float(v) for v in f.read().split()]).reshape(shape)


def read_pose(path_jpg, axis_align):
    """video_utils.py:221-227: axis_align_matrix @ pose, in f64 (the caller rounds to f32, :230)."""
    return np.asarray(axis_align, dtype=np.float64) @ read_matrix(path_jpg.replace("jpg", "txt"))


def read_rgb(path):
    """video_utils.py:286-288: Image.open(...).convert("RGB") -> uint8 [H, W, 3]."""
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def image_sizes(path_jpg):
    """((Hc, Wc), (Hd, Wd)) from the file headers of a frame's colour and depth images."""
    with Image.open(path_jpg) as im:
        wc, hc = im.size
    with Image.open(path_jpg.replace(".jpg", ".png")) as im:
        wd, hd = im.size
    return (hc, wc), (hd, wd)


def scene_layout(n, color_hw, depth_hw):
    """Byte layout of one scene's host block: name -> (shape, dtype, offset); total size."""
    lay, off = {}, 0
    for name, shape, dt in (("frames", (n, color_hw[0], color_hw[1], 3), "uint8"), ("depth", (n, depth_hw[0], depth_hw[1]), "int16"),
                            ("pose", (n, 4, 4), "float32")):
        lay[name] = (shape, dt, off)
        off += int(np.prod(shape)) * np.dtype(dt).itemsize
        off = (off + 255) // 256 * 256
    return lay, off


def views(buf, layout):
    return {k: np.ndarray(shape, dtype=dt, buffer=buf, offset=off) for k, (shape, dt, off) in layout.items()}


def decode_into(arrays, i, path_jpg, axis_align):
    """Frame i's three files -> row i of the scene's arrays; returns the seconds per stage."""
    t0 = time.perf_counter()
    arrays["depth"][i] = read_depth(path_jpg).view(np.int16)
    t1 = time.perf_counter()
    arrays["pose"][i] = read_pose(path_jpg, axis_align).astype(np.float32)
    t2 = time.perf_counter()
    arrays["frames"][i] = read_rgb(path_jpg)
    t3 = time.perf_counter()
    return {"depth_png": t1 - t0, "pose_txt": t2 - t1, "rgb_decode": t3 - t2}


# ---------------------------------------------------------------------------------------------- worker-process side
_ATTACHED = {}


def _worker_decode(shm_name, layout, i, path_jpg, axis_align):
    shm = _ATTACHED.get(shm_name)
    if shm is None:
        if len(_ATTACHED) > 64:
            for s in _ATTACHED.values():
                s.close()
            _ATTACHED.clear()
        shm = _ATTACHED[shm_name] = shared_memory.SharedMemory(name=shm_name)
        # Python < 3.13 registers an ATTACHED block with the attaching process's resource tracker, which would unlink it when this
        # worker exits; the parent owns the block
        from multiprocessing import resource_tracker
        try:
            resource_tracker.unregister(shm._name, "shared_memory")
        except Exception:
            pass
    return decode_into(views(shm.buf, layout), i, path_jpg, axis_align)


def _worker_ready(delay):
    time.sleep(delay)
    return True


def make_pool(workers):
    """A pool of `workers` decoding processes, all started NOW (forked from this process: call it before the process has a large
    heap or - cleanest - before it initialises the GPU; the children never touch the GPU, torch or anything but PIL / numpy)."""
    import concurrent.futures as cf
    import multiprocessing as mp
    pool = cf.ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("fork"))
    for f in [pool.submit(_worker_ready, 0.05) for _ in range(workers)]:       # one task per worker: forces every fork to happen here
        f.result()
    return pool
