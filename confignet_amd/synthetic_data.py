"""FFHQ-shaped synthetic datasets with the NeuralRendererDataset field contract the step functions
read (reference: confignet/neural_renderer_dataset.py:71-100,150-228): `.imgs` uint8 (M,R,R,3),
`.eye_masks` uint8 (M,R,R), `.metadata_inputs[name]` (M,in_i) + ["rotations"] (M,3),
`.metadata_input_distributions[name].sample(n) -> (values, None)`.  There is no network access for
FFHQ or the synthetic renders, so images are seeded noise (conv cost is data independent)."""
import numpy as np

from .neural_renderer_dataset import ExemplarDistribution, OneHotDistribution

# face-model input dimensionalities of the reference's test dataset
FACEMODEL_INPUT_DIMS = {
    "beard_style_embedding": 9, "blendshape_values": 62, "bone_rotations:left_eye": 3, "eye_color": 8,
    "eyebrow_style_embedding": 44, "geometry_identity_params": 53, "hdri_embedding": 50, "head_hair_color": 3,
    "head_hair_style_embedding": 18, "lower_eyelash_style": 4, "texture_embedding": 50, "upper_eyelash_style": 4,
}
ONE_HOT_INPUTS = ("eye_color", "lower_eyelash_style", "upper_eyelash_style")


class SyntheticFaceDataset:
    def __init__(self, n_imgs, res, seed=0, low_pass=True):
        rng = np.random.default_rng(seed)
        if low_pass:
            # smooth non-constant images: upsampled coarse noise + fine noise (constant images would make
            # the instance-norm / style std singular)
            coarse = rng.integers(0, 256, size=(n_imgs, res // 8, res // 8, 3)).astype(np.float32)
            img = np.repeat(np.repeat(coarse, 8, axis=1), 8, axis=2)
            img = 0.75 * img + 0.25 * rng.integers(0, 256, size=(n_imgs, res, res, 3))
            self.imgs = img.astype(np.uint8)
        else:
            self.imgs = rng.integers(0, 256, size=(n_imgs, res, res, 3), dtype=np.uint8)
        self.eye_masks = np.zeros((n_imgs, res, res), np.uint8)
        s = max(res // 12, 2)
        for i in range(n_imgs):
            y, x = rng.integers(res // 4, res // 2, size=2)
            self.eye_masks[i, y:y + s, x:x + s] = 1
        self.metadata_inputs = {}
        for name, d in FACEMODEL_INPUT_DIMS.items():
            if name in ONE_HOT_INPUTS:
                v = np.eye(d, dtype=np.float32)[rng.integers(0, d, size=n_imgs)]
            else:
                v = rng.standard_normal(size=(n_imgs, d)).astype(np.float32)
            self.metadata_inputs[name] = v
        rot = np.zeros((n_imgs, 3), np.float32)
        rot[:, 0] = np.pi * rng.uniform(-30, 30, n_imgs) / 180
        rot[:, 1] = np.pi * rng.uniform(-10, 10, n_imgs) / 180
        self.metadata_inputs["rotations"] = rot
        self.metadata_input_distributions = {}
        for n in FACEMODEL_INPUT_DIMS:                      # (neural_renderer_dataset.py:189-207: strings -> one-hot, else exemplars)
            dist = OneHotDistribution() if n in ONE_HOT_INPUTS else ExemplarDistribution()
            dist.fit(self.metadata_inputs[n])
            self.metadata_input_distributions[n] = dist

    def process_metadata(self, config, update_config=False):
        """Fills the input dimensionality of every face-model input (neural_renderer_dataset.py:150-228)."""
        if update_config:
            for name, (_, dout) in list(config["facemodel_inputs"].items()):
                if name in FACEMODEL_INPUT_DIMS:
                    config["facemodel_inputs"][name] = (FACEMODEL_INPUT_DIMS[name], dout)
        return config
