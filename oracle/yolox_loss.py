"""fp32 CPU restatement of the YOLOX loss side (TEST ORACLE).

Follows /root/reference/models/losses/yolox/yolox_loss.py and
models/layers/losses/iou_loss.py; the numbered items of SURVEY.md Appendix A
map onto the functions below.  Differences from the reference are confined to
things the reference leaves unspecified:

  * ties in the per-GT cost sort and in the IoU sort are broken by LOWEST
    anchor index (stable sort).  `torch.sort` in the reference is unstable, so
    the golden fixtures are checked tie-free at the k-th boundary;
  * the head maps are NOT overwritten in place (the reference decodes through
    a view of the caller's tensors, yolox_loss.py:210-219).

The assignment runs under no_grad; the three losses are ordinary torch
expressions, so `loss.backward()` gives the oracle gradients w.r.t. the raw
head maps.
"""
import torch
import torch.nn.functional as F


def make_grid(h, w, dtype=torch.float32):
    """yolox_loss.py:198-200 -- meshgrid(indexing='xy') of (arange(h), arange(w))
    stacked and *viewed* as (1,1,h,w,2).  For h==w this is the usual (gx, gy)
    grid; for h!=w the view scrambles it.  The quirk is reproduced verbatim."""
    xv, yv = torch.meshgrid([torch.arange(h), torch.arange(w)], indexing="xy")
    return torch.stack((xv, yv), 2).view(1, 1, h, w, 2).to(dtype).view(1, -1, 2)


def decode(head_maps, strides, num_classes):
    """yolox_loss.py:175-228.  head_maps: list of [B, 5+C, h, w] (n_anchors=1).
    Returns preds [B,A,5+C] (cx,cy,w,h in pixels; obj/cls raw logits),
    raw boxes [B,A,4], x_shifts/y_shifts/strides each [1,A]."""
    preds, raws, xs, ys, ss = [], [], [], [], []
    n_ch = num_classes + 5
    for m, s in zip(head_maps, strides):
        b, _, h, w = m.shape
        grid = make_grid(h, w, m.dtype)
        p = m.view(b, 1, n_ch, h, w).permute(0, 1, 3, 4, 2).reshape(b, h * w, n_ch)
        raws.append(p[..., :4])
        xy = (p[..., :2] + grid) * s
        wh = torch.exp(p[..., 2:4]) * s
        preds.append(torch.cat([xy, wh, p[..., 4:]], -1))
        xs.append(grid[:, :, 0])
        ys.append(grid[:, :, 1])
        ss.append(torch.full((1, h * w), float(s), dtype=m.dtype))
    return (torch.cat(preds, 1), torch.cat(raws, 1), torch.cat(xs, 1), torch.cat(ys, 1), torch.cat(ss, 1))


def eval_decode(head_maps, strides, num_classes):
    """yolox_loss.py:25-36 -- sigmoid(obj), sigmoid(cls), cxcywh -> xyxy."""
    p = decode(head_maps, strides, num_classes)[0]
    cx, cy, w, h = p[..., 0], p[..., 1], p[..., 2], p[..., 3]
    box = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1)
    return torch.cat([box, p[..., 4:].sigmoid()], -1)


def in_boxes_info(gt, strides_a, xs, ys):
    """yolox_loss.py:231-315.  gt [G,4] cxcywh; strides_a/xs/ys [A].
    Returns cand [A] bool, in_box [G,A] bool, in_ctr [G,A] bool."""
    xc = (xs * strides_a + 0.5 * strides_a)[None]
    yc = (ys * strides_a + 0.5 * strides_a)[None]
    l = (gt[:, 0] - 0.5 * gt[:, 2])[:, None]
    r = (gt[:, 0] + 0.5 * gt[:, 2])[:, None]
    t = (gt[:, 1] - 0.5 * gt[:, 3])[:, None]
    b = (gt[:, 1] + 0.5 * gt[:, 3])[:, None]
    in_box = torch.stack([xc - l, yc - t, r - xc, b - yc], 2).min(-1).values > 0.0
    rad = 2.5 * strides_a[None]
    cl, cr = gt[:, 0:1] - rad, gt[:, 0:1] + rad
    ct, cb = gt[:, 1:2] - rad, gt[:, 1:2] + rad
    in_ctr = torch.stack([xc - cl, yc - ct, cr - xc, cb - yc], 2).min(-1).values > 0.0
    cand = (in_box.sum(0) > 0) | (in_ctr.sum(0) > 0)
    return cand, in_box, in_ctr


def pairwise_iou_cxcywh(a, b):
    """iou_loss.py:391-414 with xyxy=False: `en = (tl < br)` gate, no eps."""
    tl = torch.max(a[:, None, :2] - a[:, None, 2:] / 2, b[:, :2] - b[:, 2:] / 2)
    br = torch.min(a[:, None, :2] + a[:, None, 2:] / 2, b[:, :2] + b[:, 2:] / 2)
    area_a = torch.prod(a[:, 2:], 1)
    area_b = torch.prod(b[:, 2:], 1)
    en = (tl < br).to(tl.dtype).prod(dim=2)
    area_i = torch.prod(br - tl, 2) * en
    return area_i / (area_a[:, None] + area_b - area_i)


def simota_assign_image(pred, gt_boxes, gt_cls, strides_a, xs, ys, num_classes):
    """Per-image SimOTA (yolox_loss.py:64-117 + dynamic_k_matching :318-370).

    pred [A,5+C] decoded; gt_boxes [G,4]; gt_cls [G] float.
    Returns dict with fg [A] bool, matched_gt [A] int64 (-1 on background),
    matched_iou [A] f32 (0 on background), num_fg int, and the diagnostics
    `cand`, `dynamic_k` [G], `boundary_gap` (smallest |cost_k - cost_{k+1}| /
    scale over GTs that truncate; used to certify fixtures tie-free)."""
    A = pred.shape[0]
    G = gt_boxes.shape[0]
    cand, in_box, in_ctr = in_boxes_info(gt_boxes, strides_a, xs, ys)
    idx = torch.nonzero(cand).squeeze(1)
    nc = idx.numel()
    both = (in_box & in_ctr)[:, idx]
    box_c = pred[idx, :4]
    obj_c = pred[idx, 4:5]
    cls_c = pred[idx, 5:]
    iou = pairwise_iou_cxcywh(gt_boxes, box_c)
    iou_cost = -torch.log(iou + 1e-8)
    onehot = F.one_hot(gt_cls.to(torch.int64), num_classes).float()[:, None, :].expand(G, nc, num_classes)
    p = (cls_c.float().sigmoid() * obj_c.float().sigmoid()).sqrt()[None].expand(G, nc, num_classes)
    cls_cost = F.binary_cross_entropy(p, onehot, reduction="none").sum(-1)
    cost = cls_cost + 3.0 * iou_cost + 100000.0 * (~both)

    fg = torch.zeros(A, dtype=torch.bool)
    matched_gt = torch.full((A,), -1, dtype=torch.int64)
    matched_iou = torch.zeros(A, dtype=pred.dtype)
    out = dict(cand=cand, dynamic_k=torch.zeros(G, dtype=torch.int64), boundary_gap=float("inf"))
    if nc == 0:
        # The reference would fail inside sort/min on an empty candidate set in
        # some torch versions; an image whose GTs capture no anchor contributes
        # no foreground.  (Synthetic configs guarantee >=1 candidate per GT.)
        out.update(fg=fg, matched_gt=matched_gt, matched_iou=matched_iou, num_fg=0)
        return out
    n_k = min(10, nc)
    topk = iou.sort(dim=1, descending=True, stable=True).values[:, :n_k]
    ks = torch.clamp(topk.sum(1).int(), min=1).to(torch.int64)
    out["dynamic_k"] = ks
    match = torch.zeros(G, nc)
    gap = float("inf")
    for g in range(G):
        sc, order = cost[g].sort(stable=True)
        k = int(ks[g])
        if k < nc - 1:  # yolox_loss.py:343 -- else ALL candidates are taken
            scale = max(abs(float(sc[k - 1])), 1.0)
            gap = min(gap, abs(float(sc[k]) - float(sc[k - 1])) / scale)
            order = order[:k]
        match[g, order] = 1.0
    out["boundary_gap"] = gap
    multi = match.sum(0) > 1
    if multi.any():
        amin = cost[:, multi].argmin(0)
        match[:, multi] = 0.0
        match[amin, multi] = 1.0
    fg_c = match.sum(0) > 0
    fg[idx[fg_c]] = True
    matched_gt[idx[fg_c]] = match[:, fg_c].argmax(0)
    matched_iou[idx[fg_c]] = (match * iou).sum(0)[fg_c]
    out.update(fg=fg, matched_gt=matched_gt, matched_iou=matched_iou, num_fg=int(fg_c.sum()))
    return out


def giou_loss_quirk(pred, target):
    """iou_loss.py:7-50 with loss_type='giou'.  NOTE the non-standard penalty
    (area_c - area_i)/area_c  (line 42), kept verbatim."""
    tl = torch.max(pred[:, :2] - pred[:, 2:] / 2, target[:, :2] - target[:, 2:] / 2)
    br = torch.min(pred[:, :2] + pred[:, 2:] / 2, target[:, :2] + target[:, 2:] / 2)
    area_p = torch.prod(pred[:, 2:], 1)
    area_g = torch.prod(target[:, 2:], 1)
    en = (tl < br).to(tl.dtype).prod(dim=1)
    area_i = torch.prod(br - tl, 1) * en
    iou = area_i / (area_p + area_g - area_i + 1e-16)
    c_tl = torch.min(pred[:, :2] - pred[:, 2:] / 2, target[:, :2] - target[:, 2:] / 2)
    c_br = torch.max(pred[:, :2] + pred[:, 2:] / 2, target[:, :2] + target[:, 2:] / 2)
    area_c = torch.prod(c_br - c_tl, 1)
    giou = iou - (area_c - area_i) / area_c.clamp(1e-16)
    return 1 - giou.clamp(min=-1.0, max=1.0)


def l1_targets(gt, stride, x_shifts, y_shifts, eps=1e-8):
    """get_l1_type (yolox_loss.py:373-378): the matched GT boxes [n,4] (cx,cy,w,h in pixels) expressed in the raw output
    space of their anchors (grid offsets, log of the size in stride units)."""
    return torch.stack([gt[:, 0] / stride - x_shifts, gt[:, 1] / stride - y_shifts,
                        torch.log(gt[:, 2] / stride + eps), torch.log(gt[:, 3] / stride + eps)], 1)


def yolox_loss(head_maps, labels, strides, num_classes, return_assign=False, use_l1=False):
    """Training branch of YOLOXLoss.__call__ (yolox_loss.py:38-173).

    labels [B,M,5] rows (cls,cx,cy,w,h), zero padded.  Returns the loss dict
    (same keys as the reference; `loss_l1` is the python float 0.0 unless use_l1 -- a constructor argument of the
    reference class that its plugin factory never sets, build_detection.py:137-139)."""
    preds, raws, xs, ys, ss = decode(head_maps, strides, num_classes)
    B, A, _ = preds.shape
    nlabel = (labels.sum(dim=2) > 0).sum(dim=1)
    fg_all, mg_all, mi_all = [], [], []
    num_fgs, num_gts = 0, 0
    assigns = []
    with torch.no_grad():
        for b in range(B):
            G = int(nlabel[b])
            num_gts += G
            if G == 0:
                a = dict(fg=torch.zeros(A, dtype=torch.bool), matched_gt=torch.full((A,), -1, dtype=torch.int64),
                         matched_iou=torch.zeros(A), num_fg=0, boundary_gap=float("inf"),
                         dynamic_k=torch.zeros(0, dtype=torch.int64), cand=torch.zeros(A, dtype=torch.bool))
            else:
                a = simota_assign_image(preds[b].detach(), labels[b, :G, 1:5], labels[b, :G, 0], ss[0], xs[0], ys[0], num_classes)
            assigns.append(a)
            num_fgs += a["num_fg"]
            fg_all.append(a["fg"])
            mg_all.append(a["matched_gt"])
            mi_all.append(a["matched_iou"])
    fg = torch.stack(fg_all)            # [B,A]
    mg = torch.stack(mg_all)
    mi = torch.stack(mi_all)
    n = max(num_fgs, 1)
    bidx, aidx = torch.nonzero(fg, as_tuple=True)
    gsel = mg[bidx, aidx]
    reg_t = labels[bidx, gsel, 1:5]
    cls_t = F.one_hot(labels[bidx, gsel, 0].to(torch.int64), num_classes).to(preds.dtype) * mi[bidx, aidx][:, None]
    obj_t = fg.to(preds.dtype)
    loss_iou = giou_loss_quirk(preds[bidx, aidx, :4], reg_t).sum() / n
    loss_obj = F.binary_cross_entropy_with_logits(preds[..., 4], obj_t, reduction="none").sum() / n
    loss_cls = F.binary_cross_entropy_with_logits(preds[bidx, aidx, 5:], cls_t, reduction="none").sum() / n
    # yolox_loss.py:128-135,157-160: L1 of the RAW box outputs of the foreground anchors against get_l1_type
    loss_l1 = 0.0
    if use_l1:
        loss_l1 = F.l1_loss(raws[bidx, aidx], l1_targets(reg_t, ss[0][aidx], xs[0][aidx], ys[0][aidx]), reduction="none").sum() / n
    loss = 5.0 * loss_iou + loss_obj + loss_cls + loss_l1
    out = {
        "loss": loss,
        "loss_iou": loss_iou,
        "loss_obj": loss_obj,
        "loss_cls": loss_cls,
        "loss_l1": loss_l1,
        "proportion": n / max(num_gts, 1),
    }
    if return_assign:
        out["_assign"] = dict(fg=fg, matched_gt=mg, matched_iou=mi, num_fg=num_fgs, num_gt=num_gts,
                              boundary_gap=min([a["boundary_gap"] for a in assigns] + [float("inf")]),
                              dynamic_k=[a["dynamic_k"] for a in assigns])
    return out
