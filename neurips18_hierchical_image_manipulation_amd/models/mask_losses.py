"""``MaskReconLoss`` of the box2mask path (reference ``models/mask_losses.py:12-27``) on the HIP loss kernel."""
import torch.nn as nn

from .. import ops

IGNORE_INDEX = 255


class MaskReconLoss(nn.Module):
    """``NLLLoss2d(ignore_index=255)`` of channel log-probabilities against an id map in which every position with
    ``gt_mask < 0.5`` is ignored: the mean of ``-log p[label]`` over the positions inside the mask.  The reference rewrites
    those labels to 255 on the host and calls the torch loss; here the mask is read by the kernel (``him_masked_nll_*``)."""

    def __init__(self, use_nll=True):
        super().__init__()
        assert use_nll

    def forward(self, pred_logit, gt_label_, gt_mask):
        """pred_logit (B,C,H,W) log-probabilities, gt_label_ (B,H,W) or (B,1,H,W) ids (long or float), gt_mask (B,1,H,W)."""
        label = gt_label_.float()
        if label.dim() == 3:
            label = label.unsqueeze(1)
        return ops.masked_nll(pred_logit, label.contiguous(), gt_mask)
