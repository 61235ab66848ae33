"""Import-path shim for `loss.contrastive_aud` (trainer/trainer_cavp_avs_obj.py imports ContrastLoss from there)."""
