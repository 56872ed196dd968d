"""CPU ORACLE (test infrastructure only: imported by tests/ alone, never by the product path) -- restatement of FreeAnchor.get_losses after the network forward
(basedet/models/det/free_anchor.py:38-142), torch-CPU fp32 with autograd for the gradients; the detached pieces (decode, IoU,
encode) go through the numpy box operators of oracle/box_ops.py so that every selection sees the same fp32 values as the kernels.

Parity unpinned: the reference has no test or golden vector for this model.  Two orders the reference leaves to its backend are
fixed here and in the kernels: ties of F.topk at the bag boundary (:86-88) go to the lowest anchor index, and where two gts of one
class give an anchor a box probability (:72-73, an indexed assignment with duplicate indices) the later gt's value stands.
The reference config writes FOCLA_LOSS_ALPHA (configs/det_model/freeanchor_cfg.py:10), so FOCAL_LOSS_ALPHA keeps RetinaNet's 0.25."""
import numpy as np
import torch

from . import box_ops

TINY = float(np.finfo(np.float32).tiny)


def safelog(x):
    """layers/common/function.py:35-44"""
    return torch.log(torch.clamp(x, min=TINY))


def bag_losses(logits, offsets, anchors, gt_boxes, num_gt, *, mean=(0, 0, 0, 0), std=(0.1, 0.1, 0.2, 0.2), iou_thresh=0.6,
               bucket=50, beta=0.0, reg_weight=0.75, alpha=0.25, gamma=2.0):
    """logits (N, A, K), offsets (N, A, 4): torch fp32 (requires_grad for gradients); anchors (A, 4), gt_boxes (N, G, 5),
    num_gt (N,) numpy.  Returns (pos_loss, neg_loss) as torch scalars (already weighted by alpha / 1 - alpha)."""
    N, A, K = logits.shape
    anchors = np.asarray(anchors, np.float32)
    scores = torch.sigmoid(logits)
    box_probs = []
    pos_losses = []
    eps = 1e-7
    for n in range(N):
        G = int(num_gt[n])
        info = np.asarray(gt_boxes[n][:G], np.float32)
        labels = info[:, 4].astype(np.int32) - 1
        gt = info[:, :4]
        prob = torch.zeros((A, K), dtype=torch.float32)
        if G > 0:
            pred_box = box_ops.box_decode(anchors, offsets[n].detach().numpy(), mean, std)
            ov = box_ops.box_iou(gt, pred_box)                                              # (G, A)
            t1 = np.float32(iou_thresh)
            t2 = np.clip(ov.max(axis=1, keepdims=True), np.float32(t1 + np.float32(eps)), np.float32(1.0)).astype(np.float32)
            gp = np.clip((ov - t1) / (t2 - t1), 0, 1.0).astype(np.float32)
            for g in range(G):                                                             # later gts overwrite earlier ones
                nz = np.nonzero(gp[g])[0]
                prob[torch.from_numpy(nz), int(labels[g])] = torch.from_numpy(gp[g][nz])
            # bags
            mq = box_ops.box_iou(gt, anchors)
            k = min(bucket, A)
            order = np.argsort(-mq, axis=1, kind="stable")[:, :k]                          # ties: lowest anchor index
            idx = torch.from_numpy(order.reshape(-1).astype(np.int64))
            lab = torch.from_numpy(np.repeat(labels.astype(np.int64), k))
            matched_score = scores[n][idx, lab].reshape(G, k)
            tgt = box_ops.box_encode(anchors[order.reshape(-1)], np.repeat(gt, k, axis=0), mean, std)
            d = offsets[n][idx] - torch.from_numpy(tgt.astype(np.float32))
            if beta < 1e-5:
                sl1 = d.abs()
            else:
                sl1 = torch.where(d.abs() < beta, 0.5 * d * d / beta, d.abs() - 0.5 * beta)
            reg = sl1.sum(-1) * reg_weight
            p = matched_score * torch.exp(-reg).reshape(G, k)
            w = 1.0 / (1.0 - p)
            w = w / w.sum(dim=1, keepdim=True)
            pos_losses.append(-safelog((w * p).sum(dim=1)))
        box_probs.append(prob)
    num_fg = float(np.asarray(num_gt, np.float64).sum())
    pos = (torch.cat(pos_losses).sum() if pos_losses else scores.sum() * 0) / max(1.0, num_fg)
    bp = torch.stack(box_probs, 0)
    q = scores * (1 - bp)
    neg = ((q ** gamma) * (-safelog(1.0 - q))).sum() / max(1.0, num_fg * bucket)
    return pos * alpha, neg * (1 - alpha)
