from dataclasses import dataclass, field
from typing import Optional


@dataclass
class FairseqDataclass:
    _name: Optional[str] = field(default=None, compare=False)

    @staticmethod
    def name():
        return None
