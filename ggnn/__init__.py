"""Drop-in alias: `import ggnn` resolves to the MI355X engine (same surface as the reference's
python-src/ggnn/__init__.py, which re-exports its compiled module)."""
from ggnn_amd import *  # noqa: F401,F403
from ggnn_amd import __all__, __version__  # noqa: F401
