"""Authoring-container-only stub so the reference model can be imported (never shipped to the product path)."""
class _L:
    def __getattr__(self, k):
        return lambda *a, **kw: None
logger = _L()
