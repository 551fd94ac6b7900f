"""`import deep_prior_interpolation_amd.utils as u` mirrors the reference's `import utils as u`."""
from .generic import *          # noqa: F401,F403
from .metrics import *          # noqa: F401,F403
from .torch_utils import *      # noqa: F401,F403
from .processing import *       # noqa: F401,F403
from .patch_extractor import *  # noqa: F401,F403
from .mask import *             # noqa: F401,F403
from .synthetic import *        # noqa: F401,F403
from .slopes import *          # noqa: F401,F403
from .pocs import *            # noqa: F401,F403
