"""The dataset side of the step API (reference: confignet/neural_renderer_dataset.py): the two sampling distributions
that a checkpoint's `<name>_facemodel_distr.pck` pickles, and a reader for the reference's dataset files
(`<name>.pck` + `<name>_imgs.dat`) exposing the fields the step functions read (`.imgs`, `.eye_masks`,
`.metadata_inputs`, `.metadata_input_distributions`).  Dataset CREATION (OpenFace alignment, UV-map eye masks,
Inception features; l.102-140,236-330) is out of scope."""
import os
import pickle
from collections import OrderedDict

import numpy as np

REFERENCE_MODULE = "confignet.neural_renderer_dataset"


class OneHotDistribution:
    """Uniform discrete distribution returned as one-hot rows (neural_renderer_dataset.py:22-39)."""

    def __init__(self):
        self.n_features = None

    def fit(self, X):
        self.n_features = X.shape[1]

    def sample(self, n_samples=1):
        idx = np.random.randint(0, self.n_features, size=n_samples)
        one_hot = np.zeros((n_samples, self.n_features), np.float32)
        one_hot[np.arange(n_samples), idx] = 1
        return one_hot, idx


class ExemplarDistribution:
    """Draws rows of the fitted data (neural_renderer_dataset.py:41-59); second return value is None as there."""

    def __init__(self, exemplars=None):
        self.exemplars = None
        self.n_exemplars = None
        if exemplars is not None:
            self.fit(exemplars)

    def fit(self, X):
        self.exemplars = X
        self.n_exemplars = self.exemplars.shape[0]

    def sample(self, n_samples=1):
        idx = np.random.randint(0, self.n_exemplars, size=n_samples)
        return self.exemplars[idx], None


class _RefUnpickler(pickle.Unpickler):
    """Resolves the class paths the reference pickles (`confignet.neural_renderer_dataset.*`, and the bare
    `neural_renderer_dataset.*` of older files, l.345-350) to the classes of this module."""

    def find_class(self, module, name):
        if module in (REFERENCE_MODULE, "neural_renderer_dataset", __name__) and name in _CLASSES:
            return _CLASSES[name]
        return super().find_class(module, name)


def load_pickle(path):
    with open(path, "rb") as fp:
        return _RefUnpickler(fp).load()


def dump_pickle(obj, path):
    """pickle.dump with this module's classes recorded as `confignet.neural_renderer_dataset.<Class>`."""
    import sys
    import types
    saved = {c: c.__module__ for c in _CLASSES.values()}
    created = []
    try:
        # pickle verifies that <module>.<name> IS the class: expose the classes under the reference path while dumping
        parts = REFERENCE_MODULE.split(".")
        for i in range(1, len(parts) + 1):
            modname = ".".join(parts[:i])
            if modname not in sys.modules:
                sys.modules[modname] = types.ModuleType(modname)
                created.append(modname)
        target = sys.modules[REFERENCE_MODULE]
        shadowed = {}
        for name, cls in _CLASSES.items():
            shadowed[name] = getattr(target, name, None)
            setattr(target, name, cls)
            cls.__module__ = REFERENCE_MODULE
        with open(path, "wb") as fp:
            pickle.dump(obj, fp, protocol=pickle.HIGHEST_PROTOCOL)
    finally:
        for cls, mod in saved.items():
            cls.__module__ = mod
        target = sys.modules.get(REFERENCE_MODULE)
        if target is not None:
            for name, old in shadowed.items():
                if old is None:
                    if hasattr(target, name) and REFERENCE_MODULE in created:
                        delattr(target, name)
                else:
                    setattr(target, name, old)
        for modname in created:
            sys.modules.pop(modname, None)


class NeuralRendererDataset:
    """Reader of the reference's dataset files.  Attribute names are the pickled ones (l.71-100)."""

    def __init__(self, img_shape=None, is_synthetic=True, head_rotation_range=((-30, 30), (-10, 10), (0, 0)),
                 eye_rotation_range=((-25, 25), (-15, 15), (0, 0))):
        self.img_shape = img_shape
        self.is_synthetic = is_synthetic
        self.head_rotation_range = np.array(head_rotation_range)
        self.eye_rotation_range = np.array(eye_rotation_range)
        self.imgs = None
        self.imgs_memmap_filename = None
        self.imgs_memmap_shape = None
        self.imgs_memmap_dtype = None
        self.inception_features = None
        self.render_metadata = None
        self.eye_masks = None
        self.attributes = None
        self.metadata_inputs = None
        self.metadata_input_distributions = None
        self.metadata_input_labels = None

    @staticmethod
    def load(filename):
        """neural_renderer_dataset.py:343-356: the pickle holds everything but the images, which are a raw
        uint8 memmap `<imgs_memmap_filename>` next to it."""
        dataset = load_pickle(filename)
        basedir = os.path.dirname(filename)
        dataset.imgs = np.memmap(os.path.join(basedir, dataset.imgs_memmap_filename), dataset.imgs_memmap_dtype, "r",
                                 shape=tuple(dataset.imgs_memmap_shape))
        return dataset

    def save(self, filename):
        """neural_renderer_dataset.py:332-341 (the image memmap is written separately and never pickled)."""
        imgs, self.imgs = self.imgs, None
        try:
            dump_pickle(self, filename)
        finally:
            self.imgs = imgs

    def process_metadata(self, config, update_config=False):
        """neural_renderer_dataset.py:150-228: per face-model input named in config["facemodel_inputs"] (a key, or a
        ':'-separated path into the per-image render metadata): strings -> one-hot over the sorted unique values
        (None -> "none"), lists -> float rows, dicts -> values in sorted-key order (blendshape_values gets the jaw
        opening, bone_rotations.jaw[0], appended); rotations = head bone rotation reordered [2, 0, 1]."""
        self.metadata_inputs, self.metadata_input_distributions, self.metadata_input_labels = {}, {}, {}

        def fit(data, cls):
            d = cls()
            d.fit(data)
            return d

        for input_name in config["facemodel_inputs"].keys():
            values = self.render_metadata
            for key in input_name.split(":"):
                values = [md[key] for md in values]
            values = ["none" if v is None else v for v in values]
            assert all(type(v) == type(values[0]) for v in values)
            out_dim = config["facemodel_inputs"][input_name][1]
            if isinstance(values[0], str):
                uniq, inverse = np.unique(values, return_inverse=True)
                one_hot = np.zeros((len(values), uniq.shape[0]))
                one_hot[np.arange(len(values)), inverse] = 1
                self.metadata_inputs[input_name] = one_hot
                self.metadata_input_distributions[input_name] = fit(one_hot, OneHotDistribution)
                self.metadata_input_labels[input_name] = uniq.tolist()
                n_in = int(uniq.shape[0])
            elif isinstance(values[0], list):
                assert all(len(v) == len(values[0]) for v in values)
                arr = np.array(values, dtype=np.float32)
                self.metadata_inputs[input_name] = arr
                self.metadata_input_distributions[input_name] = fit(arr, ExemplarDistribution)
                self.metadata_input_labels[input_name] = None
                n_in = arr.shape[1]
            elif isinstance(values[0], dict):
                assert all(v.keys() == values[0].keys() for v in values)
                values = [OrderedDict(sorted(v.items(), key=lambda t: t[0])) for v in values]
                self.metadata_input_labels[input_name] = list(values[0].keys())
                arr = np.array([list(v.values()) for v in values], dtype=np.float32)
                if input_name == "blendshape_values":
                    jaw = np.array([md["bone_rotations"]["jaw"][0] for md in self.render_metadata])
                    arr = np.hstack((arr, jaw[:, np.newaxis]))
                    self.metadata_input_labels[input_name].append("jaw_opening")
                self.metadata_inputs[input_name] = arr
                self.metadata_input_distributions[input_name] = fit(arr, ExemplarDistribution)
                n_in = arr.shape[1]
            else:
                raise TypeError("unsupported metadata type %s for %s" % (type(values[0]), input_name))
            if update_config:
                config["facemodel_inputs"][input_name] = (n_in, out_dim)
        head = [md["bone_rotations"]["head"] for md in self.render_metadata]
        self.metadata_inputs["rotations"] = np.array(head)[:, [2, 0, 1]]
        self.metadata_input_labels["rotations"] = None


_CLASSES = {"OneHotDistribution": OneHotDistribution, "ExemplarDistribution": ExemplarDistribution,
            "NeuralRendererDataset": NeuralRendererDataset}
