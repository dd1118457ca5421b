"""smoke(): filled in once the engine exists."""


def smoke():
    raise RuntimeError("engine not built yet")
