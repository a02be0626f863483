"""llava.video_utils on MI355X (reference: llava/video_utils.py).

Same names and call contracts as the reference: `unproject`, `VideoProcessor`
(sample_frame_files / sample_frame_files_mc / calculate_world_coords / preprocess /
process_3d_video / discrete_point), `merge_video_dict`.  File I/O (PIL depth PNG, pose txt, EmbodiedScan
pickles) stays on the host; the arithmetic runs in HBM through libv3d_hip.so:
  * unproject(...)              -> v3d_unproject_f32           (K1)
  * preprocess(...) world_coords -> v3d_unproject_sampled_u16   (K1 + nearest resize + centre crop, K2)
Tensors returned for `world_coords` live on the GPU (the eval driver's `.half().to(model.device)`,
model_scanqa.py:163-165, works on them unchanged).  cv2 is not needed.
"""
import json
import os
import pickle

import numpy as np
import torch
from PIL import Image

from v3d import frame_io, ops


def _device():
    if not torch.cuda.is_available():
        raise ops.V3DError("llava.video_utils needs an MI355X: there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def load_matrix_from_txt(path, shape=(4, 4)):
    return frame_io.read_matrix(path, shape)


def unproject(intrinsics, poses, depths):
    """(V,4,4), (V,4,4), (V,H,W) millimetres -> (V,H,W,3) f32 world coordinates (video_utils.py:38-68).
    Host tensors are uploaded and the result is brought back to where `depths` lives, so callers that
    treat it as a CPU function (preprocessing scripts) keep working."""
    dev = depths.device if depths.is_cuda else _device()
    out = ops.unproject(intrinsics.to(dev), poses.to(dev), depths.to(dev))
    return out if depths.is_cuda else out.cpu()


class VideoProcessor:
    def __init__(self, video_folder="data", annotation_dir="data/embodiedscan/", voxel_size=None, min_xyz_range=None,
                 max_xyz_range=None, frame_sampling_strategy="uniform", val_box_type="pred", metadata_dir="data/metadata"):
        self.video_folder = video_folder
        self.voxel_size = voxel_size
        self.min_xyz_range = torch.tensor(min_xyz_range) if min_xyz_range is not None else None
        self.max_xyz_range = torch.tensor(max_xyz_range) if max_xyz_range is not None else None
        self.frame_sampling_strategy = frame_sampling_strategy
        self.scene, self.scan2obj = {}, {}
        print("============frame sampling strategy: {}=============".format(frame_sampling_strategy))
        for split in ("train", "val", "test"):
            with open(os.path.join(annotation_dir, f"embodiedscan_infos_{split}.pkl"), "rb") as f:
                for item in pickle.load(f)["data_list"]:     # the dataset's own index files (trusted local data)
                    if item["sample_idx"].startswith("scannet"):
                        self.scene[item["sample_idx"]] = item
        for split in ("train", "val"):
            box_type = "gt" if split == "train" else val_box_type
            with open(os.path.join(metadata_dir, f"scannet_{split}_{box_type}_box.json")) as f:
                self.scan2obj.update(json.load(f))
        if "mc" in frame_sampling_strategy:
            with open(os.path.join(metadata_dir, "scannet_select_frames.json")) as f:
                self.mc_sampling_files = {d["video_id"]: d for d in json.load(f)}
            with open(os.path.join(metadata_dir, "pcd_discrete_0.1.pkl"), "rb") as f:
                pc_data = pickle.load(f)
            self.pc_min, self.pc_max = {}, {}
            for scene_id, pts in pc_data.items():
                arr = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
                lo = np.minimum(arr.min(0), 1000) if len(arr) else np.full(3, 1000.0)
                hi = np.maximum(arr.max(0), -1000) if len(arr) else np.full(3, -1000.0)
                self.pc_min[scene_id] = torch.tensor(lo, dtype=torch.float32) / 10
                self.pc_max[scene_id] = torch.tensor(hi, dtype=torch.float32) / 10

    # ---- a2: prefix / coverage-ratio cut of the pre-computed greedy order (video_utils.py:131-159)
    def sample_frame_files_mc(self, video_id, frames_upbound=32, do_shift=False):
        entry = self.mc_sampling_files[video_id]
        files = list(entry["frame_files"][:frames_upbound])
        gains = entry["voxel_nums"][:frames_upbound]
        ratio = 0.95 if "ratio95" in self.frame_sampling_strategy else 0.9 if "ratio90" in self.frame_sampling_strategy else 1.0
        if ratio != 1.0:
            need, covered, keep = entry["num_all_voxels"] * ratio, 0, []
            for f, n in zip(files, gains):
                keep.append(f)
                covered += n
                if covered >= need:
                    break
            files = keep
        files.sort(key=lambda p: int(p.split("/")[-1].split(".")[0]))
        return files

    # ---- a1: uniform sampling (video_utils.py:162-194)
    def sample_frame_files(self, video_id, force_sample=False, frames_upbound=0):
        files = [os.path.join(self.video_folder, img["img_path"]) for img in self.scene[video_id]["images"]]
        n = frames_upbound if force_sample else 10
        return [files[i] for i in ops.uniform_frame_indices(len(files), n)]

    # ---- host I/O (video_utils.py:214-227, 285-290) lives in v3d.frame_io (torch-free, so loader processes can run it)
    def _align(self, video_id):
        return np.array(self.scene[video_id]["axis_align_matrix"], dtype=np.float64)

    def frame_files(self, video_id, force_sample=False, frames_upbound=0):
        if "mc" in self.frame_sampling_strategy:
            return self.sample_frame_files_mc(video_id, frames_upbound, "shift" in self.frame_sampling_strategy)
        return self.sample_frame_files(video_id, force_sample, frames_upbound)

    def describe_scene(self, video_id, force_sample=False, frames_upbound=0):
        """What v3d.pipeline.AsyncSceneLoader needs to decode a scene's sampled frames off the main thread."""
        d = {"files": self.frame_files(video_id, force_sample, frames_upbound), "axis_align": self._align(video_id).tolist(),
             "K": torch.from_numpy(np.array(self.scene[video_id]["depth_cam2img"])).float()}
        if "norm" in self.frame_sampling_strategy:      # calculate_world_coords(do_normalize=True), :232-236: the pipeline clamps on the device
            scene_id = video_id.split("/")[-1]
            d["clamp"] = (self.pc_min[scene_id].tolist(), self.pc_max[scene_id].tolist())
        return d

    def load_raw(self, video_id, frame_files):
        """The sampled frames' files -> host arrays: depth [F,Hd,Wd] int16 view of the 16-bit PNG values, frames [F,Hc,Wc,3] uint8,
        pose [F,4,4] f32 (= f32(axis_align @ pose), :227-230), K [F,4,4] f32.  (The colour and depth streams differ in size: 1296 x 968
        and 640 x 480 in ScanNet.)"""
        color_hw, depth_hw = frame_io.image_sizes(frame_files[0])
        layout, nbytes = frame_io.scene_layout(len(frame_files), color_hw, depth_hw)
        buf = bytearray(nbytes)
        arrays = frame_io.views(buf, layout)
        align = self._align(video_id)
        for i, f in enumerate(frame_files):
            frame_io.decode_into(arrays, i, f, align)
        out = {k: torch.from_numpy(v) for k, v in arrays.items()}
        out["K"] = torch.from_numpy(np.array(self.scene[video_id]["depth_cam2img"])).float().unsqueeze(0).repeat(len(frame_files), 1, 1)
        return out

    def _load_depth_pose(self, video_id, frame_files):
        K = torch.from_numpy(np.array(self.scene[video_id]["depth_cam2img"])).float()
        depths = [frame_io.read_depth(p) for p in frame_files]
        poses = [frame_io.read_pose(p, self._align(video_id)) for p in frame_files]
        depth = torch.from_numpy(np.stack(depths).view(np.int16))
        pose = torch.from_numpy(np.stack(poses)).float()
        return depth, K.unsqueeze(0).repeat(len(frame_files), 1, 1), pose

    # ---- a4 + a5 (video_utils.py:196-238)
    def calculate_world_coords(self, video_id, frame_files, do_normalize=False):
        dev = _device()
        depth, K, pose = self._load_depth_pose(video_id, frame_files)
        d32 = torch.from_numpy(depth.numpy().view(np.uint16).astype(np.float32))
        wc = ops.unproject(K.to(dev), pose.to(dev), d32.to(dev))
        if do_normalize:                                           # video_utils.py:232-236
            scene_id = video_id.split("/")[-1]
            ops.clamp_xyz(wc, self.pc_min[scene_id].tolist(), self.pc_max[scene_id].tolist())
        return {"world_coords": wc}

    # ---- a4-a6 (video_utils.py:242-321); strategy "center_crop" only (what the eval drivers use)
    def preprocess(self, video_id, image_processor, force_sample=False, frames_upbound=0, strategy="center_crop", raw=None):
        """Returns the reference's dict; "images" are the F centre crops as ONE uint8 device tensor [F,crop,crop,3] - byte for byte the
        PIL crops the reference returns as a list (Pillow's bicubic resize reproduced on the device: v3d_resize_bicubic_u8) - which
        SigLipImageProcessor.preprocess takes as it takes a list of PIL images."""
        if strategy not in ("center_crop", "resize"):
            raise NotImplementedError(f"strategy {strategy!r}: the reference knows 'center_crop' and 'resize' (video_utils.py:292-306)")
        frame_files = self.frame_files(video_id, force_sample, frames_upbound)
        dev = _device()
        if raw is None:
            raw = self.load_raw(video_id, frame_files)
        crop = image_processor.crop_size["width"]
        depth, K, pose = raw["depth"].to(dev), raw["K"].to(dev), raw["pose"].to(dev)
        boundry = ops.unproject_bounds(depth, K, pose).cpu()      # over the full-resolution back-projection (:268-273)
        H, W = depth.shape[1:3]                                   # the DEPTH map's size steers the colour resize too (:269, 298-299)
        frames = raw["frames"].to(dev)
        if strategy == "resize":                                  # :293-296 (the coordinate maps go to 384 x 384 whatever the crop size)
            coords = ops.unproject_resized(depth, K, pose, 384, torch.float32)
            images = ops.resize_crop_rgb(frames, (crop, crop))
        else:
            coords = ops.unproject_sampled(depth, K, pose, crop, torch.float32)
            new_w = int(W * (crop / H))
            images = ops.resize_crop_rgb(frames, (crop, new_w), crop=(0, (new_w - crop) // 2, crop, crop))
        if "norm" in self.frame_sampling_strategy:                # calculate_world_coords(do_normalize=True): clamp to the scene's box (:232-236)
            scene_id = video_id.split("/")[-1]
            lo, hi = self.pc_min[scene_id], self.pc_max[scene_id]
            ops.clamp_xyz(coords, lo.tolist(), hi.tolist())       # (clamping commutes with the nearest-neighbour gather ...
            b = boundry.view(3, 2)                                #  ... and with min / max: the bounds of the clamped cloud)
            boundry = torch.stack([torch.minimum(torch.maximum(b[:, 0], lo), hi), torch.minimum(torch.maximum(b[:, 1], lo), hi)], 1).reshape(6)
        return {"images": images, "world_coords": coords, "video_size": images.shape[0], "boundry": boundry,
                "objects": torch.tensor(self.scan2obj[video_id])}

    def process_3d_video(self, video_id, image_processor, force_sample=False, frames_upbound=0, strategy="center_crop"):
        out = self.preprocess(video_id, image_processor, force_sample, frames_upbound, strategy)
        out["images"] = image_processor.preprocess(out["images"], return_tensors="pt")["pixel_values"]
        return out

    # ---- video_utils.py:348-358 (host list in, host list out: f32 clamp / shift / divide / round-half-even)
    def discrete_point(self, xyz):
        x = torch.tensor(xyz)
        if self.min_xyz_range is not None:
            x = torch.maximum(x, self.min_xyz_range)
        if self.max_xyz_range is not None:
            x = torch.minimum(x, self.max_xyz_range)
        if self.min_xyz_range is not None:
            x = x - self.min_xyz_range
        return (x / self.voxel_size).round().int().tolist()


def merge_video_dict(video_dict_list):
    """video_utils.py:361-373: stack per-sample dicts, collect box_input."""
    out = {"box_input": []}
    for k in video_dict_list[0]:
        if k in ("world_coords", "images", "objects"):
            out[k] = torch.stack([d[k] for d in video_dict_list])
        elif k == "box_input":
            out["box_input"].extend(d[k] for d in video_dict_list if d[k] is not None)
    out["box_input"] = torch.Tensor(out["box_input"])
    return out
