"""Authoring-container stub (never shipped to the GPU box): engine/utils.py imports cv2 at module scope."""
