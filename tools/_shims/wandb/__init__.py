"""Authoring-container stub: engine/utils.py imports wandb at module scope."""
