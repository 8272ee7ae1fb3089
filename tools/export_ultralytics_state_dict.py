"""Run ONCE on a machine that has ultralytics: turns a `.pt` detector checkpoint into the flat
safetensors state dict mangatranslator_amd loads (ultralytics pickles cannot be read without the
package).  BatchNorm may stay un-fused; it is folded at load.

    python tools/export_ultralytics_state_dict.py best.pt models/yolo/manga109-segmentation-bubble.safetensors
"""
import sys

if __name__ == "__main__":
    from safetensors.torch import save_file
    from ultralytics import YOLO
    src, dst = sys.argv[1], sys.argv[2]
    m = YOLO(src).model.float().eval()
    sd = {k: v.detach().contiguous() for k, v in m.state_dict().items() if v.dtype.is_floating_point}
    save_file(sd, dst, metadata={"names": repr(getattr(m, "names", {}))})
    print(f"wrote {len(sd)} tensors to {dst}")
