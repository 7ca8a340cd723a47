#!/usr/bin/env python
"""OUT-OF-LOOP pin of the [TF-2.1] rules the oracle hard-codes.

Run on ANY machine with TensorFlow 2.x and a checkout of microsoft/ConfigNet importable (PYTHONPATH=<reference root>; its
own dependencies cv2 / azureml are NOT needed: only confignet/dnn_models and confignet/confignet_utils.py are imported, by
file path).  It builds, with seeded weights, one Conv2dAdaIn, one Conv3dAdaIn, one DiscrBlock, transform_3d_grid_tf,
one shared-optimizer Keras-Adam trace, one InstanceNormalization, the SAME-padding cases, and records the weight NAMES and
shapes of keras.applications ResNet50 / VGG19 / VGG16 / InceptionV3 in get_weights() order (+ one seeded InceptionV3 forward), and writes

    tests/golden/tf_pins.npz

which tests/test_oracle_kat.py::test_oracle_matches_tensorflow_pins consumes when present (skipped otherwise).  Nothing in
this repository can run it (TensorFlow is not installable here); it is shipped so that whoever has TF can turn the
"parity unpinned" label of DESIGN.md section 7 into a pinned one by committing the file it writes.

    python scripts/tf_pin_dump.py --reference /path/to/ConfigNet --out tests/golden/tf_pins.npz
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="root of the microsoft/ConfigNet checkout")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tf_pins.npz"))
    ap.add_argument("--no-applications", action="store_true", help="skip keras.applications (needs the imagenet weight download)")
    args = ap.parse_args()
    import tensorflow as tf
    from tensorflow import keras

    # the reference's package __init__ imports cv2 / azureml: load only the three modules needed, under a stub package
    pkg = types.ModuleType("confignet")
    pkg.__path__ = [os.path.join(args.reference, "confignet")]
    sys.modules["confignet"] = pkg
    dnn = types.ModuleType("confignet.dnn_models")
    dnn.__path__ = [os.path.join(args.reference, "confignet", "dnn_models")]
    sys.modules["confignet.dnn_models"] = dnn
    inorm = load_by_path("confignet.dnn_models.instance_normalization", os.path.join(args.reference, "confignet", "dnn_models", "instance_normalization.py"))
    # confignet_utils.py imports azure_ml_utils and matplotlib at module top: provide a stub for the former
    sys.modules["confignet.azure_ml_utils"] = types.ModuleType("confignet.azure_ml_utils")
    pkg.azure_ml_utils = sys.modules["confignet.azure_ml_utils"]
    utils = load_by_path("confignet.confignet_utils", os.path.join(args.reference, "confignet", "confignet_utils.py"))
    pkg.confignet_utils = utils
    bb = load_by_path("confignet.dnn_models.building_blocks", os.path.join(args.reference, "confignet", "dnn_models", "building_blocks.py"))

    rng = np.random.default_rng(0)
    out = {"tf_version": np.array(tf.__version__)}

    def seeded(model):
        ws = [rng.standard_normal(w.shape).astype(np.float32) * (0.3 if w.ndim > 1 else 0.1) for w in model.get_weights()]
        model.set_weights(ws)
        return ws

    def pack(prefix, arrays):
        out[prefix + "_n"] = np.array(len(arrays))
        for i, a in enumerate(arrays):
            out["%s_%d" % (prefix, i)] = np.asarray(a)

    # --- Conv2dAdaIn / Conv3dAdaIn (building_blocks.py:11-80): conv(same) -> LeakyReLU() -> AdaIn --------------------
    mlp_nl = lambda: keras.layers.LeakyReLU(alpha=0.2)      # hologan_generator.py:21
    for name, cls, xshape, kernel in (("conv2d_adain", bb.Conv2dAdaIn, (2, 9, 8, 6), 4), ("conv3d_adain", bb.Conv3dAdaIn, (2, 5, 4, 6, 3), 3)):
        x = rng.standard_normal(xshape).astype(np.float32)
        z = rng.standard_normal((2, 7)).astype(np.float32)
        layer = cls(num_feature_maps=8, kernel_size=kernel, double_conv=False, non_linear_after=None, z_size=7, mlp_num_units=5,
                    mlp_num_layers=2, mlp_non_linear=mlp_nl)
        layer({"x": x, "z": z})                                # builds the weights
        ws = seeded(layer)
        y = layer({"x": x, "z": z})
        out[name + "_x"], out[name + "_z"], out[name + "_y"] = x, z, y.numpy()
        pack(name + "_w", ws)
        out[name + "_weight_names"] = np.array([w.name for w in layer.weights])
    # --- DiscrBlock (building_blocks.py:83-111) -----------------------------------------------------------------------------
    for tag, hw in (("even", (8, 10)), ("odd", (9, 7))):
        x = rng.standard_normal((2, hw[0], hw[1], 5)).astype(np.float32)
        blk = bb.DiscrBlock(num_feature_maps=6, kernel_size=3, return_styles=True)
        blk(x)
        ws = seeded(blk)
        y, style = blk(x)
        out["discr_block_%s_x" % tag], out["discr_block_%s_y" % tag], out["discr_block_%s_style" % tag] = x, y.numpy(), style.numpy()
        pack("discr_block_%s_w" % tag, ws)
    # --- transform_3d_grid_tf + euler_angles_to_matrix (confignet_utils.py:63-145) with gradients ----------------------------
    grid = tf.Variable(rng.standard_normal((2, 16, 16, 16, 3)).astype(np.float32))
    ang = tf.Variable(np.array([[0.3, -0.1, 0.05], [-0.45, 0.15, 0.0]], np.float32))
    cot = rng.standard_normal((2, 16, 16, 16, 3)).astype(np.float32)
    with tf.GradientTape() as tape:
        R = utils.euler_angles_to_matrix(ang)
        o = utils.transform_3d_grid_tf(grid, R)
        s = tf.reduce_sum(o * cot)
    g_grid, g_ang = tape.gradient(s, [grid, ang])
    out.update(rot_grid=grid.numpy(), rot_angles=ang.numpy(), rot_matrix=R.numpy(), rot_out=o.numpy(), rot_cot=cot,
               rot_g_grid=g_grid.numpy(), rot_g_angles=g_ang.numpy())
    # --- InstanceNormalization(axis=-1) and get_layer_style --------------------------------------------------------------------
    x = rng.standard_normal((2, 6, 5, 4)).astype(np.float32)
    inl = inorm.InstanceNormalization(axis=-1)
    inl(x)
    ws = seeded(inl)
    out["inorm_x"], out["inorm_y"] = x, inl(x).numpy()
    pack("inorm_w", ws)
    mu, sd = utils.get_layer_style(tf.constant(x))
    out["style_mean"], out["style_std"] = mu.numpy(), sd.numpy()
    # --- Keras Adam: ONE optimizer applied to three variables in turn, twice (confignet_first_stage.py:601-610) ------------------
    opt = keras.optimizers.Adam(lr=4e-4, beta_1=0.0, beta_2=0.9)
    vs = [tf.Variable(rng.standard_normal(50).astype(np.float32)) for _ in range(3)]
    gs = [rng.standard_normal(50).astype(np.float32) for _ in range(6)]
    out["adam_theta0"] = np.stack([v.numpy() for v in vs])
    out["adam_grads"] = np.stack(gs)
    trace = []
    for it in range(2):
        for j, v in enumerate(vs):
            opt.apply_gradients([(tf.constant(gs[3 * it + j]), v)])
            trace.append(v.numpy().copy())
    out["adam_trace"] = np.stack(trace)
    out["adam_iterations"] = np.array(int(opt.iterations.numpy()))
    opt2 = keras.optimizers.Adam(lr=1e-4)                                  # fine_tune_on_img's optimizer (defaults)
    v = tf.Variable(out["adam_theta0"][0])
    t2 = []
    for it in range(3):
        opt2.apply_gradients([(tf.constant(gs[it]), v)])
        t2.append(v.numpy().copy())
    out["adam_default_trace"] = np.stack(t2)
    # --- SAME padding: a delta image through Conv2D with a ones kernel shows the (lo, hi) split ----------------------------------
    for k, s, n in ((4, 1, 8), (3, 2, 8), (3, 2, 9), (3, 1, 8)):
        conv = keras.layers.Conv2D(1, k, strides=s, padding="same", use_bias=False, kernel_initializer="ones")
        x = np.zeros((1, n, n, 1), np.float32)
        x[0, 0, 0, 0] = 1.0
        x[0, n - 1, n - 1, 0] = 2.0
        out["same_k%d_s%d_n%d" % (k, s, n)] = conv(x).numpy()
    # --- LeakyReLU defaults --------------------------------------------------------------------------------------------------------
    out["keras_leakyrelu_of_minus1"] = keras.layers.LeakyReLU()(tf.constant([-1.0])).numpy()
    out["tf_nn_leaky_relu_of_minus1"] = tf.nn.leaky_relu(tf.constant([-1.0])).numpy()
    out["layernorm_default_eps"] = np.array(keras.layers.LayerNormalization().epsilon)
    # --- whole networks of the reference: get_weights() ORDER (shapes) and one seeded forward each (R11: Keras orders a subclassed
    # model's weights by attribute-tracking order, which the checkpoint format model.npz relies on) ---------------------------------
    try:
        gen_mod = load_by_path("confignet.dnn_models.hologan_generator", os.path.join(args.reference, "confignet", "dnn_models", "hologan_generator.py"))
        disc_mod = load_by_path("confignet.dnn_models.hologan_discriminator", os.path.join(args.reference, "confignet", "dnn_models", "hologan_discriminator.py"))
        L = 9
        gen = gen_mod.HologanGenerator(latent_dim=L, output_shape=(128, 128), n_adain_mlp_units=8, n_adain_mlp_layers=2, gen_output_activation="tanh")
        z = rng.standard_normal((1, L)).astype(np.float32)
        rot = np.array([[0.2, -0.1, 0.0]], np.float32)
        inp = gen.build_input_dict(z, rot)
        gen(inp)
        ws = seeded(gen)
        out["generator_weight_shapes"] = np.array([str(tuple(w.shape)) for w in ws])
        out["generator_z"], out["generator_rot"], out["generator_img"] = z, rot, gen(inp).numpy()
        pack("generator_w", ws)
        d = disc_mod.HologanDiscriminator(img_shape=(64, 64), num_resample=5, disc_max_feature_maps=512, disc_kernel_size=3,
                                          disc_expansion_factor=48, initial_from_rgb_layer_in_discr=True)
        x = rng.uniform(-1, 1, (2, 64, 64, 3)).astype(np.float32)
        d(x)
        ws = seeded(d)
        o = d(x)
        out["discriminator_weight_shapes"] = np.array([str(tuple(w.shape)) for w in ws])
        out["discriminator_x"] = x
        out["discriminator_out_keys"] = np.array(list(o.keys()))
        out["discriminator_out"] = np.concatenate([np.asarray(v).reshape(2, 1) for v in o.values()], axis=1)
        pack("discriminator_w", ws)
    except Exception as e:                                     # (an older / newer reference layout: keep the operator-level pins)
        out["network_pins_error"] = np.array(repr(e))
    # --- keras.applications: get_weights() order (names + shapes) ------------------------------------------------------------------
    if not args.no_applications:
        for name, ctor in (("resnet50", lambda: keras.applications.ResNet50(weights=None, include_top=False, input_shape=(224, 224, 3), pooling="avg")),
                           ("vgg19", lambda: keras.applications.VGG19(weights=None, include_top=False, input_shape=(224, 224, 3))),
                           ("vgg16", lambda: keras.applications.VGG16(weights=None, include_top=False, input_shape=(224, 224, 3))),
                           # the FID / KID feature extractor (confignet/metrics/inception_distance.py:12)
                           ("inception_v3", lambda: keras.applications.InceptionV3(weights=None, include_top=False, input_shape=(256, 256, 3), pooling="avg"))):
            m = ctor()
            out[name + "_weight_names"] = np.array([w.name for w in m.weights])
            out[name + "_weight_shapes"] = np.array([str(tuple(w.shape)) for w in m.weights])
            out[name + "_layer_names"] = np.array([l.name for l in m.layers])
        # BatchNormalization epsilon of ResNet50 and the preprocess_input constants
        out["resnet50_bn_eps"] = np.array([l.epsilon for l in keras.applications.ResNet50(weights=None, include_top=False, input_shape=(64, 64, 3)).layers
                                           if isinstance(l, keras.layers.BatchNormalization)][:1])
        # one seeded forward of InceptionV3 (weights = the seeded arrays written next to it, in get_weights() order)
        inc = keras.applications.InceptionV3(weights=None, include_top=False, input_shape=(139, 107, 3), pooling="avg")
        rs = np.random.RandomState(7)
        ws = [(rs.uniform(0.5, 1.5, size=w.shape) if "moving_variance" in v.name else rs.normal(size=w.shape) * (0.05 if w.ndim == 4 else 0.1)).astype(np.float32)
              for w, v in zip(inc.get_weights(), inc.weights)]
        inc.set_weights(ws)
        xin = rs.uniform(-1, 1, size=(2, 139, 107, 3)).astype(np.float32)
        out["inception_v3_probe_input"] = xin
        out["inception_v3_probe_seed"] = np.array(7)
        out["inception_v3_probe_features"] = inc.predict(xin)
        out["inception_preprocess_probe"] = keras.applications.inception_v3.preprocess_input(np.array([[0.0, 127.5, 255.0]], np.float32))
        probe = np.zeros((1, 2, 2, 3), np.float32)
        probe[..., 0], probe[..., 1], probe[..., 2] = 10.0, 20.0, 30.0
        out["caffe_preprocess_probe"] = keras.applications.resnet50.preprocess_input(probe.copy())
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, "with", len(out), "arrays (tensorflow %s)" % tf.__version__)


if __name__ == "__main__":
    main()
