"""Authoring-container stub: engine/utils.py imports AsciiTable at module scope."""


class AsciiTable:
    def __init__(self, *a, **k):
        self.table = ""
