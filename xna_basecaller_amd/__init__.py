"""
xna_basecaller_amd -- MI355X (gfx950) implementation of the CSB5/XNA_Basecaller `bonito basecaller`
hot path behind the reference's plugin surface.

A model directory's config.toml names this package (or keeps `bonito.crf`, which load_symbol maps
here) under [model] package; `Model` and `basecall` are then resolved exactly as the reference does
(ub-bonito/bonito/util.py:228-239, cli/basecaller.py:61).
"""
__version__ = "0.1.0"

from .crf import Model, basecall  # noqa: F401,E402
