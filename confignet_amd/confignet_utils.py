"""Config / bookkeeping helpers (reference: confignet/confignet_utils.py:14-61,198-212)."""
import copy
import json
import sys

import numpy as np


def merge_configs(default_config, input_config):
    """Overlay `input_config` on `default_config` (semantics of confignet_utils.py:39-61): a key whose default is a dict is
    merged key by key (the override must be a dict too), every other key present in the input replaces the default -- also when
    the input holds a dict where the default holds a scalar --, and keys only the input has are carried over.  The result never
    aliases the defaults' nested dicts (callers mutate their config)."""
    merged = copy.deepcopy(dict(default_config))
    for key, override in input_config.items():
        base = default_config.get(key)
        if isinstance(base, dict):
            assert isinstance(override, dict), "config key %r: a dict default needs a dict override" % (key,)
            merged[key] = merge_configs(base, override)
        else:
            merged[key] = override
    return merged


def load_confignet(model_path):
    """Dispatch on config["model_type"] (confignet_utils.py:14-21)."""
    with open(model_path, "r") as fp:
        metadata = json.load(fp)
    cls = getattr(sys.modules["confignet_amd"], metadata["model_type"])
    return cls.load(model_path)


def flip_random_subset_of_images(images):
    """Mirror a random half of the batch left-right, in place (confignet_utils.py:198-204).  The draw -- one
    np.random.randint(0, 2) per image from the global NumPy stream -- is part of the parity contract; the training path takes
    the same flags and applies them on device while it gathers the batch (neural_renderer_dataset)."""
    chosen = np.flatnonzero(np.random.randint(0, 2, size=images.shape[0]))
    images[chosen] = images[chosen, :, ::-1]
    return images


def update_loss_dict(main_loss_dict, new_loss_dict):
    """Append this step's scalars to the per-loss histories (confignet_utils.py:206-212)."""
    for name in new_loss_dict:
        main_loss_dict.setdefault(name, []).append(float(new_loss_dict[name]))


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
