"""Drop-in for `bonito.crf` (ub-bonito/bonito/crf/__init__.py): exports Model and basecall."""
from .model import Model  # noqa: F401
from .basecall import basecall  # noqa: F401
