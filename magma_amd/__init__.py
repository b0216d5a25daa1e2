"""magma_amd -- MI355X-native MAGMA (CLIP RN50x16 -> ImagePrefix -> GPT-J-6B +
adapters) behind the reference's Python surface.  Hot-path arithmetic lives in
libmagma_hip.so (hand-written gfx950 kernels, C ABI in include/magma_hip.h)."""
from .config import MultimodalConfig
from .image_input import ImageInput
from .language_model import get_gptj
from .magma import Magma
from .transforms import get_transforms
from .utils import (configure_param_groups, count_parameters, cycle, get_tokenizer, is_main, load_model,
                    parse_args, print_main, save_model)

__all__ = ["Magma", "ImageInput", "MultimodalConfig", "get_gptj", "get_transforms", "configure_param_groups",
           "count_parameters", "cycle", "get_tokenizer", "is_main", "load_model", "parse_args", "print_main",
           "save_model"]
