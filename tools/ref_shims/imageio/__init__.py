"""Import-time stub (imageio is absent offline); visualisation is out of scope."""


class v3:  # noqa: N801
    pass
