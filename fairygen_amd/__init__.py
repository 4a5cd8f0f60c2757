"""fairygen_amd — MI355X-native implementation of FairyGen's animation inference hot path
(diffsynth WanVideoPipeline -> Wan2.2-TI2V-5B denoise loop -> Wan2.2 VAE decode).

Same call surface as the reference's ``diffsynth.pipelines.wan_video`` for that path; compute runs on
hand-written gfx950 HIP kernels behind the C ABI of ``include/fairygen_hip.h`` (``fairygen_amd.hip``).
"""
from .loader import ModelConfig, ModelPool, hash_model_file, load_state_dict  # noqa: F401
from .flow_match import FlowMatchScheduler  # noqa: F401
from .lora import GeneralLoRALoader  # noqa: F401
from .wan_video import WanVideoPipeline, model_fn_wan_video  # noqa: F401
from .data import save_video  # noqa: F401
