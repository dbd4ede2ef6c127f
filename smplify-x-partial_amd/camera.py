"""Drop-in for smplifyx/camera.py:35-117: `create_camera(camera_type='persp', ...)` ->
`PerspectiveCamera` with Parameters rotation[B,3,3], translation[B,3] and buffers center[B,2],
focal_length_x/y[B].  Inside the fitting loop the projection (and its adjoint) lives in the HIP
closure kernel; this module is the state holder the caller mutates in place
(fit_single_frame.py:399-401,409-411,469-470) plus a stand-alone forward for callers that
project points themselves."""
import torch
import torch.nn as nn


def create_camera(camera_type="persp", **kwargs):
    if camera_type.lower() == "persp":
        return PerspectiveCamera(**kwargs)
    raise ValueError("Uknown camera type: {}".format(camera_type))


class PerspectiveCamera(nn.Module):
    FOCAL_LENGTH = 5000

    def __init__(self, rotation=None, translation=None, focal_length_x=None, focal_length_y=None, batch_size=1,
                 center=None, dtype=torch.float32, **kwargs):
        super().__init__()
        self.batch_size, self.dtype = batch_size, dtype
        self.register_buffer("zero", torch.zeros([batch_size], dtype=dtype))

        def focal(v):
            if v is None or type(v) == float:
                return torch.full([batch_size], self.FOCAL_LENGTH if v is None else v, dtype=dtype)
            return v
        self.register_buffer("focal_length_x", focal(focal_length_x))
        self.register_buffer("focal_length_y", focal(focal_length_y))
        self.register_buffer("center", torch.zeros([batch_size, 2], dtype=dtype) if center is None else center)
        if rotation is None:
            rotation = torch.eye(3, dtype=dtype).unsqueeze(0).repeat(batch_size, 1, 1)
        self.rotation = nn.Parameter(rotation, requires_grad=True)
        if translation is None:
            translation = torch.zeros([batch_size, 3], dtype=dtype)
        self.translation = nn.Parameter(translation, requires_grad=True)

    def forward(self, points):
        """uv = f * (R p + t)_xy / (R p + t)_z + c, no z clamp (camera.py:93-117)."""
        cam = torch.einsum("bki,bji->bjk", self.rotation, points) + self.translation.unsqueeze(1)
        img = cam[:, :, :2] / cam[:, :, 2:3]
        f = torch.stack([self.focal_length_x, self.focal_length_y], dim=-1).unsqueeze(1)
        return img * f + self.center.unsqueeze(1)
