"""CPU restatement of the evaluation formatting step (TEST ORACLE -- imported by tests only).

`format_outputs` follows /root/reference/models/evaluators/postprocess.py:95-138 box by box, with
`xyxy2xywh` from /root/reference/models/utils/bbox.py:58-63.  The reference MODULE cannot be imported
here (its first line imports torchvision, which is not installable in the build container), but the
reference FUNCTION runs: tools/gen_golden.py parses the reference file, executes only `format_outputs` (with
the reference's own xyxy2xywh) and records tests/golden/format_outputs.npz -- this restatement is PINNED to
that fixture (tests/test_oracle_nms.py::test_format_outputs_vs_reference_fixture), records, per-class arrays
and the in-place rescale side effect included.
"""
import numpy as np
import torch


def xyxy2xywh(bboxes):
    """bbox.py:58-63: [x1, y1, x2, y2] -> [x1, y1, w, h] on a copy."""
    y = bboxes.clone()
    y[:, 2] = bboxes[:, 2] - bboxes[:, 0]
    y[:, 3] = bboxes[:, 3] - bboxes[:, 1]
    return y


def format_outputs(outputs, ids, hws, val_size, class_ids, labels=None):
    """postprocess.py:95-138.  `outputs`: list of CPU fp32 tensors [n, 6] (x1, y1, x2, y2, conf, cls) or None.
    Returns (json_list, det_list); rescales the boxes of `outputs[i]` in place like the reference (:112-113)."""
    json_list = []
    det_list = [[np.empty(shape=[0, 5]) for _ in range(len(class_ids))] for _ in range(len(outputs))]
    for i, (output, img_h, img_w, img_id) in enumerate(zip(outputs, hws[0], hws[1], ids)):
        if output is None:
            continue
        bboxes = output[:, 0:4]
        scale = min(val_size[0] / float(img_w), val_size[1] / float(img_h))
        bboxes /= scale
        coco_bboxes = xyxy2xywh(bboxes)
        scores = output[:, 4]
        clses = output[:, 5]
        for bbox, cocobox, score, cls in zip(bboxes, coco_bboxes, scores, clses):
            cls = int(cls)
            json_list.append({
                "image_id": int(img_id),
                "category_id": class_ids[cls],
                "bbox": cocobox.numpy().tolist(),
                "score": score.numpy().item(),
                "segmentation": [],
            })
        for c in range(len(class_ids)):
            det_list[i][c] = output[clses == c, 0:5].numpy()
    return json_list, det_list
