"""Eval post-processing with the reference's call signature
(reference models/evaluators/postprocess.py:7-48):

    postprocess(predictions, conf_thre=0.7, nms_thre=0.45, class_agnostic=False)
        -> list (one entry per image) of Tensor[n, 6] (x1, y1, x2, y2, conf, cls) or None

`predictions` is the [B, A, 5+C] tensor OneStageD returns in eval mode.  The whole
batch is handled by one device launch sequence (csrc/nms.hip): class max, confidence
filter, ordered compaction (first 10 000 in anchor order), stable score sort,
torchvision-rule batched NMS, first 300 -- followed by ONE device->host copy of the
per-image counts to cut the ragged result list (the reference syncs once per image).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import NmsDesc, call, PlyoloError

MAX_DET = 300     # postprocess.py:8
MAX_NMS = 10000   # postprocess.py:9
NUMEL_THRESHOLD = 20000  # torchvision batched_nms: coordinate trick up to 20 000 box coordinates on GPU

_ws_cache = {}


def postprocess_device(predictions, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    """Device-resident result: (det [B, 300, 6], count int32[B]) -- no host sync."""
    if not predictions.is_cuda:
        raise PlyoloError("postprocess runs on an MI355X device tensor; there is no CPU path")
    if predictions.dim() != 3 or predictions.shape[2] < 6:
        raise PlyoloError("predictions must be [B, A, 5+C]")
    p = predictions.contiguous().float()
    B, A, nch = p.shape
    d = NmsDesc()
    d.B, d.A, d.C, d.conf_thre, d.nms_thre, d.class_agnostic = B, A, nch - 5, float(conf_thre), float(nms_thre), int(bool(class_agnostic))
    d.max_nms, d.max_det, d.numel_threshold = MAX_NMS, MAX_DET, NUMEL_THRESHOLD
    key = (B, A, nch, p.device)
    ws = _ws_cache.get(key)
    if ws is None:
        nbytes = _lib.lib().plyolo_postprocess_workspace(C.byref(d))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
        _ws_cache.clear()
        _ws_cache[key] = ws
    det = torch.empty(B, MAX_DET, 6, dtype=torch.float32, device=p.device)
    count = torch.empty(B, dtype=torch.int32, device=p.device)
    call("plyolo_postprocess", C.byref(d), p.data_ptr(), det.data_ptr(), count.data_ptr(), None, ws.data_ptr(), ws.numel(),
         torch.cuda.current_stream().cuda_stream)
    return det, count


def postprocess(predictions, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    if predictions.shape[0] == 0:
        return []
    det, count = postprocess_device(predictions, conf_thre, nms_thre, class_agnostic)
    counts = count.tolist()  # the only host sync
    return [det[i, :n] if n > 0 else None for i, n in enumerate(counts)]


def format_outputs(outputs, ids, hws, val_size, class_ids, labels=None):
    """Detections -> (COCO json records, per-image per-class VOC arrays), the reference's
    `format_outputs(outputs, ids, hws, val_size, class_ids, labels)`
    (reference models/evaluators/postprocess.py:95-138, models/utils/bbox.py:58-63; called from
    PL_Modules/pl_detection.py:78,132 right after `postprocess`).

    Same results and the same side effect (the boxes of `outputs[i]` are rescaled IN PLACE to the original
    image, postprocess.py:112-113).  The rescale and the xyxy -> xywh conversion of the whole batch are ONE HIP launch
    (plyolo_format_detections) followed by ONE device->host copy; the reference copies every box, score and class mask
    to the host separately (postprocess.py:125-126,136), i.e. thousands of blocking copies per validation batch.
    """
    import ctypes as C
    import numpy as np
    from ._lib import FmtImage
    n_cls = len(class_ids)
    det_list = [[np.empty(shape=[0, 5]) for _ in range(n_cls)] for _ in range(len(outputs))]
    live, row0 = [], 0
    for i, (output, img_h, img_w, img_id) in enumerate(zip(outputs, hws[0], hws[1], ids)):
        if output is None:
            continue
        if not output.is_cuda:
            raise PlyoloError("format_outputs runs on MI355X device tensors (the output of postprocess); there is no CPU path")
        if output.dtype != torch.float32 or output.dim() != 2 or output.shape[1] < 6 or output.stride(1) != 1:
            raise PlyoloError("format_outputs expects fp32 [n, 6] detection rows (x1, y1, x2, y2, conf, cls)")
        scale = min(val_size[0] / float(img_w), val_size[1] / float(img_h))
        live.append((i, int(img_id), output, float(scale), row0))
        row0 += int(output.shape[0])
    if not live or row0 == 0:
        return [], det_list
    dev = live[0][2].device
    arr = (FmtImage * len(live))()
    for k, (i, img_id, output, scale, r0) in enumerate(live):
        arr[k].det, arr[k].n, arr[k].ld, arr[k].row0, arr[k].scale = output.data_ptr(), int(output.shape[0]), int(output.stride(0)), r0, scale
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    packed = torch.empty(row0, 8, dtype=torch.float32, device=dev)
    call("plyolo_format_detections", table.data_ptr(), len(live), max(int(o.shape[0]) for _, _, o, _, _ in live), packed.data_ptr(),
         torch.cuda.current_stream().cuda_stream)
    host = packed.cpu().numpy()  # the only device->host copy (x1, y1, x2, y2, w, h, score, cls)
    json_list = []
    for i, img_id, output, scale, r0 in live:
        n = int(output.shape[0])
        d = host[r0:r0 + n]
        clses = d[:, 7]
        boxes = d[:, [0, 1, 4, 5]].tolist()
        scores = d[:, 6].tolist()
        for k in range(n):
            json_list.append({
                "image_id": img_id,
                "category_id": class_ids[int(clses[k])],
                "bbox": boxes[k],
                "score": scores[k],
                "segmentation": [],
            })
        xyxys = d[:, [0, 1, 2, 3, 6]]
        for c in range(n_cls):
            det_list[i][c] = xyxys[clses == c].copy()
    return json_list, det_list
