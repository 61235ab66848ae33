"""Authoring-container stub: engine/utils.py does `import torchvision.transforms as transforms` at module scope."""
