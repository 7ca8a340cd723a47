"""Config / bookkeeping helpers (reference: confignet/confignet_utils.py:14-61,198-212)."""
import copy
import json
import sys

import numpy as np


def merge_configs(default_config, input_config):
    """Recursively merge configuration dictionaries (confignet_utils.py:39-61)."""
    result = {}
    for name in default_config:
        lhs = default_config[name]
        if name in input_config:
            rhs = input_config[name]
            if isinstance(lhs, dict):
                assert isinstance(rhs, dict)
                result[name] = merge_configs(lhs, rhs)
            else:
                result[name] = rhs
        else:
            result[name] = copy.deepcopy(lhs)      # never alias the defaults' nested dicts: callers mutate their config
    for name in input_config:
        rhs = input_config[name]
        if isinstance(rhs, dict) and name in default_config.keys():
            continue
        result[name] = rhs
    return result


def load_confignet(model_path):
    """Dispatch on config["model_type"] (confignet_utils.py:14-21)."""
    with open(model_path, "r") as fp:
        metadata = json.load(fp)
    cls = getattr(sys.modules["confignet_amd"], metadata["model_type"])
    return cls.load(model_path)


def flip_random_subset_of_images(images):
    """Host version kept for API parity (confignet_utils.py:198-204); the training path draws the same
    flip flags and applies them on device while gathering the batch."""
    flip_or_not = np.random.randint(0, 2, size=images.shape[0])
    for i, flip in enumerate(flip_or_not):
        if flip == 1:
            images[i] = np.fliplr(images[i])
    return images


def update_loss_dict(main_loss_dict, new_loss_dict):
    """confignet_utils.py:206-212."""
    for key, val in new_loss_dict.items():
        val = float(val)
        main_loss_dict.setdefault(key, []).append(val)


def log_loss_vals(loss_dict, output_dir, step_number, prefix, aml_run=None, tb_log_writer=None):
    """confignet_utils.py:214-241 without the matplotlib / TensorBoard sinks: <prefix>losses.txt, one column per loss."""
    import os
    os.makedirs(output_dir, exist_ok=True)
    loss_names, loss_vals = list(loss_dict.keys()), list(loss_dict.values())
    if not loss_vals:
        return
    if aml_run is not None:
        from . import azure_ml_utils
        azure_ml_utils.log_losses(aml_run, loss_names, [x[-1] for x in loss_vals], prefix)
    np.savetxt(os.path.join(output_dir, prefix + "losses.txt"), np.stack(loss_vals, axis=1), header="\t".join(loss_names))


def write_image(path, bgr_image):
    """cv2.imwrite stand-in (OpenCV is not a dependency here): PNG / JPEG through Pillow when it is installed, else .npy."""
    try:
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(bgr_image[..., ::-1])).save(path)
    except ImportError:
        np.save(path + ".npy", bgr_image)


def build_image_matrix(images, n_rows, n_cols):
    """confignet_utils.py:182-190: tiles images[j * n_cols + i] into an (n_rows*H, n_cols*W, 3) uint8 canvas."""
    h, w = images.shape[1:3]
    out = np.zeros((n_rows * h, n_cols * w, 3), dtype=np.uint8)
    for i in range(n_cols):
        for j in range(n_rows):
            out[j * h:(j + 1) * h, i * w:(i + 1) * w] = images[j * n_cols + i]
    return out
