"""tests/golden/frames/<clip>_16x224.npz: the reference's example clips as the context encoder receives them.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs /root/reference):

    python oracle/make_golden_frames.py [davis_camel panda]

BASELINE.json configs[1] / configs[3] name `assets/examples/davis_camel` and the panda clip.  The frames are DATA (16 RGBA PNGs of
512 x 512 with the background already removed: `BackgroundRemover._has_a_valid_alpha_mask` is true for them, so the reference skips
RMBG); what is stored is what the reference hands to DINOv2:
  1. the reference's OWN `ImagePreprocessor.process_images` (actionmesh/preprocessing/image_processor.py:124-150: composite on white,
     shared bounding box over the clip, square padding 10 %) - imported unmodified from /root/reference;
  2. the geometric half of `BitImageProcessor.preprocess` (image_encoder.py:48-51) with the DINOv2 preprocessor settings
     (facebook/dinov2-large preprocessor_config.json: shortest edge 256 bicubic, centre crop 224) - transformers' own class;
as uint8 RGB (16, 224, 224, 3).  The remaining arithmetic - rescale by 1/255, normalise by the ImageNet mean / std - is applied by
`frames_to_pixels` below where the fixture is read (tools/e2e_synthetic.py --clip), so the stored bytes stay an image.
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def frames_to_pixels(rgb_u8):
    """(T, 224, 224, 3) uint8 -> (T, 3, 224, 224) float32 pixel_values: BitImageProcessor's do_rescale + do_normalize."""
    import torch
    x = torch.from_numpy(np.asarray(rgb_u8)).float() * (1.0 / 255.0)
    x = (x - torch.tensor(MEAN)) / torch.tensor(STD)
    return x.permute(0, 3, 1, 2).contiguous()


if __name__ == "__main__":
    import importlib.util
    from PIL import Image
    # the reference's module file, unmodified, loaded by path: its package __init__ also imports background_removal.py, which needs
    # cv2 (absent offline) - and these clips never reach RMBG
    spec = importlib.util.spec_from_file_location("ref_image_processor", "/root/reference/actionmesh/preprocessing/image_processor.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ImagePreprocessor = mod.ImagePreprocessor                                     # reference
    from transformers import BitImageProcessor
    proc = BitImageProcessor(do_resize=True, size={"shortest_edge": 256}, resample=3, do_center_crop=True, crop_size={"height": 224, "width": 224},
                             do_rescale=False, do_normalize=False, do_convert_rgb=True)
    for clip in (sys.argv[1:] or ["davis_camel", "panda"]):
        files = sorted(glob.glob(f"/root/reference/assets/examples/{clip}/*.png"))
        frames = [Image.open(f) for f in files]
        frames = ImagePreprocessor().process_images(frames)
        px = proc.preprocess(frames, return_tensors="np").pixel_values            # (T, 3, 224, 224), values 0 .. 255
        rgb = np.clip(np.rint(np.asarray(px)), 0, 255).astype(np.uint8).transpose(0, 2, 3, 1)
        path = os.path.join(ROOT, "tests", "golden", "frames", f"{clip}_16x224.npz")
        np.savez_compressed(path, rgb_u8=rgb, source=np.array(f"assets/examples/{clip} ({len(files)} frames)"))
        print("wrote", path, rgb.shape, os.path.getsize(path), "bytes; mean", float(rgb.mean()))
