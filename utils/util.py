from mage_amd.utils.util import *  # noqa: F401,F403
from mage_amd.utils.util import instantiate_from_config, get_obj_from_str, default, zero_module, exists  # noqa: F401
