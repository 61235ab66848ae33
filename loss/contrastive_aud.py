"""`loss.contrastive_aud.ContrastLoss` of the reference (loss/contrastive_aud.py:8-141), served by the MI355X implementation:
same constructor `ContrastLoss(temperature, ignore_idx, max_views)` and `forward(embeds_match, gt_match, embeds_shuffle,
gt_shuffle)`."""
from cavp_amd.contrast import ContrastLoss  # noqa: F401

__all__ = ["ContrastLoss"]
