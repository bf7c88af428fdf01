"""gad - MI355X-native (gfx950) hot path of the Shapley data-attribution engine.

The names exported here are the ones the reference takes from `diffusers`
(main.py:22-24, src/diffusion_utils.py:15-23): entry points import this module in place of
diffusers for the objects on the hot path."""
from . import _capi, ops  # noqa: F401
from .ops import operand_precision, set_operand_precision  # noqa: F401
from .nn import LoRALinearLayer, UNet2DModel  # noqa: F401
from .pipelines import DDIMPipeline, DDPMPipeline, LDMPipeline, StableDiffusionLatentPipeline, sd_simple_loss  # noqa: F401
from .schedulers import DDIMScheduler, DDPMScheduler  # noqa: F401
from .training import EMAModel, FusedTrainer, flatten_params, lr_lambda  # noqa: F401
from . import coalition  # noqa: F401,E402
from .coalition import DeviceLoader, antithetic_timesteps, seed_everything  # noqa: F401,E402

__version__ = "0.1.0"
from .scoring import diversity_against_dataset, fid_against_dataset, global_scores_against_dataset  # noqa: F401,E402
from .sd import UNet2DConditionModel  # noqa: F401,E402
