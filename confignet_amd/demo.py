"""Headless driver of the demo inference loop (reference: evaluation/confignet_demo.py:154-201 + evaluation/basic_ui.py):
per frame, interpolate towards the target embeddings, splice the eye-gaze latents (one tiny MLP predict), run the smoothed /
fine-tuned generator at N <= 6 and tile the images.  The window / keyboard of the reference (cv2) is replaced by `key()`;
`run(test_mode=True)` fires every key handler once like the reference's --test_mode (basic_ui.py:69-127).  The generator
forward of a frame is ONE replayed HIP graph (ConfigNetFirstStage._replay_generator)."""
import numpy as np

from .confignet_utils import build_image_matrix

STEP = 0.05       # rotation_angle_step_size (basic_ui.py:23)


class DemoSession:
    def __init__(self, confignet_model, latentgan_model=None, input_images=None, n_rows=2, n_cols=3, hdri_turntable_embeddings=None):
        self.model, self.latentgan, self.input_images = confignet_model, latentgan_model, input_images
        self.n_rows, self.n_cols = n_rows, n_cols
        self.exit = False
        self.rotation_offset = np.zeros((1, 3))
        self.eye_rotation_offset = np.zeros((1, 3))
        self.controlled_param_idx = 0
        self.facemodel_param_names = [n for n in confignet_model.config["facemodel_inputs"].keys() if n != "bone_rotations:left_eye"]
        self.interp_0 = self.interp_1 = None
        self.interpolation_coef, self.n_interpolation_steps = 1.0, 5
        self.hdri_turntable_embeddings = hdri_turntable_embeddings     # assets/hdri_turntable_embeddings.npy in the reference
        self.current_hdri_frame, self.sweeping_hdri = 0, False
        self.embedding_unmodified, self.rotation, self.orig_images = self.get_new_embeddings()
        self.set_next_embeddings(self.embedding_unmodified)

    # ---- confignet_demo.py:62-84 ----
    def get_new_embeddings(self):
        if self.input_images is None:
            n = self.n_rows * self.n_cols
            emb = self.latentgan.generate_latents(n, truncation=0.7)
            rot = np.zeros((n, 3), dtype=np.float32)
            return emb, rot, self.model.generate_images(emb, rot)
        if len(self.input_images) == 1:
            self.n_rows = self.n_cols = 1
        n = self.n_rows * self.n_cols
        idx = np.random.randint(0, len(self.input_images), n)
        orig = np.array([self.input_images[i] for i in idx])
        emb, rot = self.model.encode_images(orig)
        return emb, rot, orig

    # ---- basic_ui.py:35-59 ----
    def set_next_embeddings(self, embeddings):
        self.interp_0 = embeddings if self.interp_0 is None else self.current_frame_embeddings()
        self.interp_1 = embeddings
        self.interpolation_coef = 0

    def current_frame_embeddings(self):
        emb = self.interp_0 * (1 - self.interpolation_coef) + self.interp_1 * self.interpolation_coef
        if self.sweeping_hdri and self.hdri_turntable_embeddings is not None:
            emb = self.model.set_facemodel_param_in_latents(emb, "hdri_embedding", self.hdri_turntable_embeddings[self.current_hdri_frame])
            self.current_hdri_frame = (self.current_hdri_frame + 1) % len(self.hdri_turntable_embeddings)
        return emb

    def frame(self):
        """One rendered frame (confignet_demo.py:154-165): returns the (n_rows*R, n_cols*(2R+20), 3) uint8 canvas."""
        emb = self.current_frame_embeddings()
        emb = self.model.set_facemodel_param_in_latents(emb, "bone_rotations:left_eye", self.eye_rotation_offset)
        gen = self.model.generate_images(emb, self.rotation + self.rotation_offset)
        strip = np.full((gen.shape[0], gen.shape[1], 20, 3), 255, np.uint8)
        canvas = build_image_matrix(np.dstack((self.orig_images, gen, strip)), self.n_rows, self.n_cols)
        if self.interpolation_coef < 1.0:                                   # perform_per_frame_actions
            self.interpolation_coef = min(self.interpolation_coef + 1.0 / self.n_interpolation_steps, 1.0)
        return canvas

    def key(self, k, test_mode=False):
        """The key handlers of basic_ui.py:69-127 and confignet_demo.py:175-201 (k: one character, or 27 for Esc)."""
        k = chr(k) if isinstance(k, int) and k != 27 else k
        k = k.lower() if isinstance(k, str) else k
        on = lambda c: test_mode or k == c
        if k == 27 or test_mode:
            self.exit = True
        for c, vec, axis, sign in (("a", self.rotation_offset, 0, -1), ("d", self.rotation_offset, 0, 1), ("w", self.rotation_offset, 1, -1),
                                   ("s", self.rotation_offset, 1, 1), ("q", self.rotation_offset, 2, -1), ("e", self.rotation_offset, 2, 1),
                                   ("j", self.eye_rotation_offset, 2, -1), ("l", self.eye_rotation_offset, 2, 1),
                                   ("i", self.eye_rotation_offset, 0, -1), ("k", self.eye_rotation_offset, 0, 1),
                                   ("u", self.eye_rotation_offset, 1, -1), ("o", self.eye_rotation_offset, 1, 1)):
            if on(c):
                vec[0, axis] += sign * STEP
        if on("z"):
            self.controlled_param_idx = (self.controlled_param_idx - 1) % len(self.facemodel_param_names)
        if on("c"):
            self.controlled_param_idx = (self.controlled_param_idx + 1) % len(self.facemodel_param_names)
        if on("n"):
            self.sweeping_hdri = not self.sweeping_hdri
        if on(" "):
            self.embedding_unmodified, self.rotation, self.orig_images = self.get_new_embeddings()
            self.set_next_embeddings(self.embedding_unmodified)
        if on("v"):
            self.set_next_embeddings(self.embedding_unmodified)
        if on("x"):
            name = self.facemodel_param_names[self.controlled_param_idx]
            value = self.model.facemodel_param_distributions[name].sample(1)[0]
            self.set_next_embeddings(self.model.set_facemodel_param_in_latents(self.current_frame_embeddings(), name, value))
        if on("b"):
            if self.input_images is None or len(self.input_images) != 1:
                print("For one-shot learning to work you need to specify a single input image path")
            else:
                self.embedding_unmodified, self.rotation = self.model.fine_tune_on_img(self.input_images[0], 1 if test_mode else 50)
                self.set_next_embeddings(self.embedding_unmodified)
        return k

    def run(self, test_mode=False, keys=(), on_frame=None):
        """The demo loop: one frame per key of `keys` (or a single frame with every handler fired in test mode)."""
        frames = 0
        it = iter(keys)
        while not self.exit:
            canvas = self.frame()
            frames += 1
            if on_frame is not None:
                on_frame(canvas)
            if test_mode:
                self.key(None, test_mode=True)
                break
            k = next(it, 27)
            self.key(k)
        return frames
