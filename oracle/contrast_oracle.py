"""CPU oracle of the reference's live contrastive loss (loss/contrastive_aud.py::ContrastLoss) - TEST INFRASTRUCTURE.

Plain-PyTorch restatement; consumes `torch.randperm` from the default CPU generator in the same order as the reference
(per kept foreground class in ascending order, then background, then shuffle candidates), so for an identical RNG state
it samples the identical anchors.  Pinned by tests/golden/contrast.npz (output + input gradients of the reference class
itself, tools/make_golden.py)."""
import torch
import torch.nn.functional as F


def contrast_loss(embeds_match, gt_match, embeds_shuffle, gt_shuffle, temperature=0.1, ignore_idx=255, max_views=512,
                  eps=1e-12):
    """contrastive_aud.py:17-37."""
    size = embeds_match.shape[2:]
    gm = F.interpolate(gt_match.unsqueeze(1).float(), size=size, mode="nearest").squeeze(1).long()
    gs = F.interpolate(gt_shuffle.unsqueeze(1).float(), size=size, mode="nearest").squeeze(1).long()
    em = F.normalize(embeds_match, p=2, dim=1).flatten(2).permute(0, 2, 1)
    es = F.normalize(embeds_shuffle, p=2, dim=1).flatten(2).permute(0, 2, 1)
    gm, gs = gm.flatten(1), gs.flatten(1)
    fg = (gm > 0) & (gm != ignore_idx)                      # :97-98
    e_fg, g_fg = em[fg], gm[fg]
    chosen_e, chosen_g = [], []
    for item in torch.unique(g_fg):                         # :76-89
        cur = g_fg == item
        if int(cur.sum()) < max_views:
            continue
        r = torch.randperm(int(cur.sum()))
        chosen_g.append(g_fg[cur][r][:max_views])
        chosen_e.append(e_fg[cur][r][:max_views])
    if not chosen_e:
        return torch.tensor([0.0])
    e_bg, g_bg = em[gm == 0], gm[gm == 0]                   # :109-110
    e_sh, g_sh = es[fg], gs[fg]                             # :114-115
    k = int(min(max_views, e_sh.shape[0], e_bg.shape[0]))   # :118
    i1, i2 = torch.randperm(e_bg.shape[0]), torch.randperm(e_sh.shape[0])
    anchors = torch.cat(chosen_e + [e_bg[i1][:k], e_sh[i2][:k]], 0)
    labels = torch.cat(chosen_g + [g_bg[i1][:k], g_sh[i2][:k]], 0)
    return info_nce(anchors, labels, temperature, eps)


def info_nce(anchors, labels, temperature, eps=1e-12):
    """contrastive_aud.py:41-74 with contras_ = anchors.clone()."""
    lab = labels.unsqueeze(1)
    mask = torch.eq(lab, lab.t()).float()
    logits = anchors @ anchors.clone().t() / temperature
    logits = logits - logits.max(dim=1, keepdim=True)[0].detach()
    neg_mask = 1 - mask
    mask = mask.clone().fill_diagonal_(0.0)
    neg = (torch.exp(logits) * neg_mask).sum(1, keepdim=True)
    log_prob = logits - torch.log(torch.exp(logits) + neg)
    mlpp = (mask * log_prob).sum(1) / (mask.sum(1) + eps)
    return -mlpp.mean()
