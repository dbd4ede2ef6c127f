"""Drop-in for smplifyx/data_parser.py:44-281 -- the dataset side of the hot path: OpenPose-style
keypoint json -> [persons, K, 3] arrays in the column order the joint mapper expects, plus the
per-joint optimisation weights.  Same factory, class and method names; images are read with PIL
(the reference uses cv2, absent here) and only serve H, W downstream.
"""
import json
import os
import os.path as osp
from collections import namedtuple
from glob import glob

import numpy as np
import torch

from .utils import smpl_to_annotation

Keypoints = namedtuple("Keypoints", ["keypoints", "gender_gt", "gender_pd"])
Keypoints.__new__.__defaults__ = (None,) * len(Keypoints._fields)


def create_dataset(format="coco25", data_folder="data", **kwargs):
    if format.lower() == "coco25":
        return COCO25(data_folder, **kwargs)
    if format.lower() == "halpe":
        return Halpe(data_folder, **kwargs)
    if format.lower() == "coco_wholebody":
        return COCO_Wholebody(data_folder, **kwargs)
    raise ValueError("Unknown dataset: {}".format(format))


def read_keypoints(keypoint_fn, use_hands=True, use_face=True, use_face_contour=False):
    """data_parser.py:57-109: body | left hand | right hand | 51 inner face | 17 contour."""
    with open(keypoint_fn) as fh:
        data = json.load(fh)
    keypoints, gender_pd, gender_gt = [], [], []
    arr = lambda person, key: np.array(person[key], dtype=np.float32).reshape([-1, 3])
    for person in data["people"]:
        parts = [arr(person, "pose_keypoints_2d")]
        if use_hands:
            parts += [arr(person, "hand_left_keypoints_2d"), arr(person, "hand_right_keypoints_2d")]
        if use_face:
            face = arr(person, "face_keypoints_2d")
            parts.append(face[17:17 + 51])
            if use_face_contour:
                parts.append(face[:17])
        if "gender_pd" in person:
            gender_pd.append(person["gender_pd"])
        if "gender_gt" in person:
            gender_gt.append(person["gender_gt"])
        keypoints.append(np.concatenate(parts, axis=0))
    return Keypoints(keypoints=keypoints, gender_pd=gender_pd, gender_gt=gender_gt)


def read_image(img_path):
    """float32 RGB in [0, 1], [H, W, 3] (cv2.imread(...)[:, :, ::-1] / 255 in the reference)."""
    from PIL import Image
    with Image.open(img_path) as im:
        return np.asarray(im.convert("RGB"), np.float32) / 255.0


class COCO25(object):
    """data_parser.py:112-227 (a torch Dataset there; plain iterable here)."""

    def __init__(self, data_folder, img_folder="images", keyp_folder="keypoints", use_hands=False, use_face=False,
                 dtype=torch.float32, model_type="smplx", joints_to_ign=None, use_face_contour=False,
                 format="coco25", num_body_joints=25, num_hand_joints=20, **kwargs):
        self.use_hands, self.use_face, self.use_face_contour = use_hands, use_face, use_face_contour
        self.model_type, self.dtype, self.joints_to_ign, self.format = model_type, dtype, joints_to_ign, format
        self.num_body_joints, self.num_hand_joints = num_body_joints, num_hand_joints
        self.num_joints = self.num_body_joints + 2 * self.num_hand_joints * use_hands
        self.img_folder = osp.join(data_folder, img_folder)
        self.keyp_folder = osp.join(data_folder, keyp_folder)
        # operator precedence kept from the reference (:144-147): every .png, and every .jpg not starting with '.'
        self.img_paths = sorted(osp.join(self.img_folder, fn) for fn in os.listdir(self.img_folder)
                                if fn.endswith(".png") or fn.endswith(".jpg") and not fn.startswith("."))
        self.cnt = 0

    def get_model2data(self):
        return smpl_to_annotation(self.model_type, use_hands=self.use_hands, use_face=self.use_face,
                                  use_face_contour=self.use_face_contour, format=self.format)

    def get_left_shoulder(self):
        return 2

    def get_right_shoulder(self):
        return 5

    def get_joint_weights(self):
        w = np.ones(self.num_joints + 2 * self.use_hands + self.use_face * 51 + 17 * self.use_face_contour, np.float32)
        if self.joints_to_ign is not None and -1 not in self.joints_to_ign:
            w[self.joints_to_ign] = 0.0
        return torch.tensor(w, dtype=self.dtype)

    def __len__(self):
        return len(self.img_paths)

    def __getitem__(self, idx):
        return self.read_item(self.img_paths[idx])

    def keypoint_file(self, img_path):
        img_fn, _ = osp.splitext(osp.split(img_path)[1])
        fns = glob(osp.join(self.keyp_folder, img_fn + "_*.json"))
        if len(fns) == 0:
            raise Exception("Keypoint file for {} does not exist!".format(img_fn))
        return img_fn, fns[0]

    def read_item(self, img_path, with_image=True):
        img_fn, keypoint_fn = self.keypoint_file(img_path)
        kt = read_keypoints(keypoint_fn, use_hands=self.use_hands, use_face=self.use_face,
                            use_face_contour=self.use_face_contour)
        if len(kt.keypoints) < 1:
            return {}
        out = {"fn": img_fn, "img_path": img_path, "keypoints": np.stack(kt.keypoints)}
        if with_image:
            out["img"] = read_image(img_path)
        if kt.gender_gt:
            out["gender_gt"] = kt.gender_gt
        if kt.gender_pd:
            out["gender_pd"] = kt.gender_pd
        return out

    def __iter__(self):
        return self

    def __next__(self):
        return self.next()

    def next(self):
        if self.cnt >= len(self.img_paths):
            raise StopIteration
        self.cnt += 1
        return self.read_item(self.img_paths[self.cnt - 1])


class Halpe(COCO25):
    def __init__(self, data_folder, format="halpe", **kwargs):
        kwargs.pop("num_body_joints", None); kwargs.pop("num_hand_joints", None)
        super(Halpe, self).__init__(data_folder, format=format, num_body_joints=26, num_hand_joints=20, **kwargs)

    def get_left_shoulder(self):
        return 5

    def get_right_shoulder(self):
        return 6


class COCO_Wholebody(COCO25):
    def __init__(self, data_folder, format="coco_wholebody", **kwargs):
        kwargs.pop("num_body_joints", None); kwargs.pop("num_hand_joints", None)
        super(COCO_Wholebody, self).__init__(data_folder, format=format, num_body_joints=23, num_hand_joints=20,
                                             **kwargs)

    def get_left_shoulder(self):
        return 5

    def get_right_shoulder(self):
        return 6
