"""FreeAnchor (basedet/models/det/free_anchor.py) on the HIP path: the RetinaNet network with bag losses instead of the
IoU matcher.  Everything up to the logits / offsets and everything behind d_logits / d_offsets is RetinaNet's; the loss is one
C-ABI call (bd_freeanchor_loss_fwd_bwd, csrc/freeanchor.hip)."""
import torch

from .. import ops
from ..utils.registry import registers
from .retinanet import RetinaNet


@registers.models.register()
class FreeAnchor(RetinaNet):
    def _plan_head(self, pl):
        super()._plan_head(pl)
        pl.fa_ws = None

    def get_losses(self, inputs):
        """FreeAnchor.get_losses (free_anchor.py:20-142): {"total_loss", "pos_loss", "neg_loss"}."""
        assert self.training
        pre = self.pre_process(inputs)
        pl = pre["plan"]
        self._cur = pl
        self.network_forward(pl)
        m = self.cfg.MODEL
        gt = pre["gt_boxes"]
        num_gt = pre["img_info"][:, 4].to(torch.int32).contiguous()
        N, Gmax = gt.shape[0], gt.shape[1]
        bucket = m.BUCKET.BUCKET_SIZE
        need = ops.freeanchor_workspace_bytes(N, Gmax, bucket, pl.A_total)
        if pl.fa_ws is None or pl.fa_ws.numel() < need:
            pl.fa_ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        ops.freeanchor_loss_fwd_bwd(pl.logits, pl.offsets, self.box_ld, self.num_anchors, pl.anchors, self.num_classes, gt, num_gt,
                                    m.BOX_REG.MEAN, m.BOX_REG.STD, m.BUCKET.BOX_IOU_THRESH, bucket, m.LOSSES.SMOOTH_L1_BETA,
                                    m.LOSSES.REG_LOSS_WEIGHT, m.LOSSES.FOCAL_LOSS_ALPHA, m.LOSSES.FOCAL_LOSS_GAMMA, pl.loss_buf,
                                    pl.d_logits, pl.d_offsets, pl.fa_ws)
        pos_loss, neg_loss = pl.loss_buf[0], pl.loss_buf[1]
        return {"total_loss": pos_loss + neg_loss, "pos_loss": pos_loss, "neg_loss": neg_loss}
