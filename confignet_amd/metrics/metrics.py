"""InceptionMetrics (reference: confignet/metrics/metrics.py:201-264): KID / FID of generated images against the inception
features of a fixed sample of the training set, logged per metrics checkpoint."""
import os

import numpy as np

from .inception_distance import InceptionFeatureExtractor, compute_FID, compute_KID


class InceptionMetrics:
    def __init__(self, confignet_config, dataset, n_samples_for_metrics=1000, weights_path=None):
        self.n_samples_for_metrics = n_samples_for_metrics
        self._config, self._dataset = confignet_config, dataset
        self._weights_path = weights_path or confignet_config.get("inception_weights_path")
        self._metric_sample_idxs = np.random.randint(0, dataset.imgs.shape[0], n_samples_for_metrics)    # metrics.py:206
        # The reference builds the extractor and slices dataset.inception_features here; both are deferred to the first
        # metrics checkpoint (same values; a run that never reaches one -- benchmarks, step tests -- does not pay for an
        # InceptionV3 and 1000 feature vectors).  The np.random draw above stays where the reference has it.
        self._extractor = None
        self._gt_features = None

    @property
    def inception_feature_extractor(self):
        if self._extractor is None:
            self._extractor = InceptionFeatureExtractor(self._config["output_shape"], self._weights_path)
        return self._extractor

    @property
    def gt_inception_features(self):
        if self._gt_features is None:
            feats = getattr(self._dataset, "inception_features", None)
            if feats is not None and self._weights_path is None:
                # the dataset's precomputed features come from the imagenet InceptionV3; without its weights the extractor
                # here is a seeded-random network, and FID / KID between features of two DIFFERENT networks mean nothing:
                # recompute the ground-truth side with the extractor that sees the generated images
                feats = None
            if feats is None:
                # the reference's dataset files carry precomputed features (neural_renderer_dataset.py:323-325); a dataset
                # without them gets the features of the sampled images computed here, with this extractor
                imgs = self._dataset.imgs
                import torch
                if torch.is_tensor(imgs):
                    sel = imgs[torch.as_tensor(self._metric_sample_idxs, device=imgs.device)].cpu().numpy()
                else:
                    sel = np.asarray(imgs)[self._metric_sample_idxs]
                self._gt_features = self.inception_feature_extractor.get_features(sel)
            else:
                self._gt_features = np.asarray(feats)[self._metric_sample_idxs]
        return self._gt_features

    def get_metrics(self, generated_images):
        generated_inception_features = self.inception_feature_extractor.get_features(generated_images)
        kid = compute_KID(generated_inception_features, self.gt_inception_features)
        fid = compute_FID(generated_inception_features, self.gt_inception_features)
        return kid, fid

    def update_and_log_metrics(self, images, metrics_dict, output_dir, aml_run=None, tb_log_writer=None):
        """metrics.py:216-264 without the matplotlib / TensorBoard / AzureML sinks: appends to metrics_dict["kid" | "fid"]
        and rewrites <output_dir>/inception_metrics.txt (step_number, kid, fid per row, the reference's header)."""
        os.makedirs(output_dir, exist_ok=True)
        kid, fid = self.get_metrics(images)
        metrics_dict.setdefault("kid", []).append(kid)
        metrics_dict.setdefault("fid", []).append(fid)
        assert len(metrics_dict["kid"]) == len(metrics_dict["fid"])
        if "training_step_number" in metrics_dict:
            steps = metrics_dict["training_step_number"]
            assert len(steps) == len(metrics_dict["kid"])
        else:
            steps = range(len(metrics_dict["kid"]))
        if aml_run is not None:
            aml_run.log("Kernel Inception Distance", kid)
            aml_run.log("Frechet Inception Distance", fid)
        rows = np.stack((list(steps), metrics_dict["kid"], metrics_dict["fid"]), axis=1)
        np.savetxt(os.path.join(output_dir, "inception_metrics.txt"), rows, delimiter="\t", header="\t".join(["step_number", "kid", "fid"]))
        return kid, fid
