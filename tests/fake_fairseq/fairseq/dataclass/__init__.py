from .configs import FairseqDataclass  # noqa: F401
