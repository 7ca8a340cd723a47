"""Data formats either side of the hot path, CPU only.

tests/golden/reference_assets/test_dataset_res_256.{pck,_imgs.dat} are the dataset files the reference's own tests hold
(reference tests/test_assets, used by tests/training_test.py:13-23): data, read here by confignet_amd's reader."""
import copy
import os
import pickletools

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSET = os.path.join(ROOT, "tests", "golden", "reference_assets", "test_dataset_res_256.pck")


def test_reference_dataset_file_loads_and_process_metadata_matches_the_known_dimensions():
    from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
    from confignet_amd.neural_renderer_dataset import ExemplarDistribution, NeuralRendererDataset, OneHotDistribution
    ds = NeuralRendererDataset.load(ASSET)
    assert ds.imgs.shape == (2, 256, 256, 3) and ds.imgs.dtype == np.uint8 and ds.is_synthetic
    assert ds.eye_masks.shape == (2, 256, 256) and list(ds.eye_masks.sum(axis=(1, 2))) == [0, 369]
    assert ds.inception_features.shape == (2, 2048) and len(ds.attributes) == 2 and len(ds.attributes[0]) == 40
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    ds.process_metadata(cfg, True)
    # input dimensionalities of the reference's face model (from render_metadata[0])
    want = {"texture_embedding": 50, "geometry_identity_params": 53, "blendshape_values": 62, "beard_style_embedding": 9,
            "eyebrow_style_embedding": 44, "head_hair_style_embedding": 18, "hdri_embedding": 50, "head_hair_color": 3,
            "bone_rotations:left_eye": 3}
    for k, n in want.items():
        assert cfg["facemodel_inputs"][k][0] == n and ds.metadata_inputs[k].shape == (2, n), k
    for k in ("eye_color", "lower_eyelash_style", "upper_eyelash_style"):                               # strings -> one-hot
        x = ds.metadata_inputs[k]
        assert x.shape[0] == 2 and np.all(x.sum(axis=1) == 1) and isinstance(ds.metadata_input_distributions[k], OneHotDistribution)
        assert cfg["facemodel_inputs"][k][0] == x.shape[1] == len(ds.metadata_input_labels[k])
    assert ds.metadata_input_labels["blendshape_values"][-1] == "jaw_opening"
    assert isinstance(ds.metadata_input_distributions["texture_embedding"], ExemplarDistribution)
    # rotations: head bone rotation reordered [2, 0, 1] (neural_renderer_dataset.py:224-226)
    head = np.array([md["bone_rotations"]["head"] for md in ds.render_metadata])
    assert np.array_equal(ds.metadata_inputs["rotations"], head[:, [2, 0, 1]])
    np.random.seed(0)
    v, idx = ds.metadata_input_distributions["texture_embedding"].sample(5)
    assert v.shape == (5, 50) and idx is None
    v, idx = ds.metadata_input_distributions["eye_color"].sample(4)
    assert v.shape[0] == 4 and np.array_equal(v.argmax(axis=1), idx)


def test_distribution_pickles_carry_the_reference_class_paths(tmp_path):
    """<name>_facemodel_distr.pck must be loadable by the reference and vice versa: classes are recorded as
    confignet.neural_renderer_dataset.<Class> (confignet_first_stage.py:177-180,200-204)."""
    from confignet_amd import neural_renderer_dataset as nrd
    d = {"a": nrd.ExemplarDistribution(np.arange(12, dtype=np.float32).reshape(4, 3)), "b": nrd.OneHotDistribution()}
    d["b"].fit(np.zeros((3, 5)))
    path = str(tmp_path / "m_facemodel_distr.pck")
    nrd.dump_pickle(d, path)
    strings = [arg for op, arg, _ in pickletools.genops(open(path, "rb").read()) if isinstance(arg, str)]
    assert "confignet.neural_renderer_dataset" in strings and "confignet_amd.neural_renderer_dataset" not in strings
    assert nrd.ExemplarDistribution.__module__ == "confignet_amd.neural_renderer_dataset"             # restored
    back = nrd.load_pickle(path)
    assert isinstance(back["a"], nrd.ExemplarDistribution) and back["a"].n_exemplars == 4 and back["b"].n_features == 5
    # the alias package resolves the same path with plain pickle as well (what the reference does)
    import pickle
    import confignet   # noqa: F401
    with open(path, "rb") as fp:
        again = pickle.load(fp)
    assert type(again["a"]) is nrd.ExemplarDistribution


def test_confignet_alias_exports_what_the_reference_scripts_import():
    import confignet
    from confignet.confignet_first_stage import DEFAULT_CONFIG
    from confignet.latent_gan import DEFAULT_CONFIG as LG
    assert DEFAULT_CONFIG["optimizer"]["lr"] == 0.0004 and LG["optimizer"]["lr"] == 0.00005
    for name in ("ConfigNetFirstStage", "ConfigNet", "LatentGAN", "NeuralRendererDataset", "load_confignet"):
        assert hasattr(confignet, name), name
    assert confignet.azure_ml_utils.get_aml_run() is None
    assert confignet.confignet_utils.merge_configs({"a": {"b": 1, "c": 2}}, {"a": {"b": 3}}) == {"a": {"b": 3, "c": 2}}
    import train_confignet
    import train_latent_gan   # noqa: F401
    assert [f for f, _ in train_confignet.FLAGS][:11] == [
        "--output_dir", "--log_dir", "--data_dir", "--real_training_set_path", "--synth_training_set_path", "--validation_set_path",
        "--attribute_classifier_path", "--batch_size", "--stage_1_training_steps", "--stage_2_training_steps", "--n_samples_for_metrics"]
