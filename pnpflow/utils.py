from pnpflow_amd.utils import *  # noqa: F401,F403
from pnpflow_amd.utils import (CfgNode, load_cfg_from_cfg_file, merge_cfg_from_list, define_model, load_model, postprocess,  # noqa: F401
                               compute_psnr, compute_average_psnr, get_save_path_ip, save_time_use)
