"""Authoring-container-only stub of the timm==0.4.9 symbols the reference imports."""
