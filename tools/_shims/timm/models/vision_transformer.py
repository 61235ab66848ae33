def _cfg(url="", **kwargs):
    return dict(url=url, **kwargs)
