"""``LiDARPoints`` — the slice of the reference's point structure the GGA train pipeline touches
(mmdet3d/core/points/base_points.py:11-440, lidar_points.py): ``tensor [N, points_dim]``,
``coord``, ``in_range_3d`` (strict inequalities, :203-225), ``shuffle`` (``torch.randperm``,
:135-143), ``cat`` (:356-377), indexing, ``new_point``."""
import numpy as np
import torch


class BasePoints:
    def __init__(self, tensor, points_dim=3, attribute_dims=None):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device('cpu')
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, points_dim)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == points_dim, tensor.size()
        self.tensor = tensor
        self.points_dim = points_dim
        self.attribute_dims = attribute_dims
        self.rotation_axis = 0

    @property
    def coord(self):
        return self.tensor[:, :3]

    @property
    def shape(self):
        return self.tensor.shape

    @property
    def bev(self):
        return self.tensor[:, [0, 1]]

    @property
    def device(self):
        return self.tensor.device

    def shuffle(self):
        idx = torch.randperm(len(self), device=self.tensor.device)
        self.tensor = self.tensor[idx]
        return idx

    def in_range_3d(self, point_range):
        t = self.tensor
        return ((t[:, 0] > point_range[0]) & (t[:, 1] > point_range[1]) & (t[:, 2] > point_range[2])
                & (t[:, 0] < point_range[3]) & (t[:, 1] < point_range[4]) & (t[:, 2] < point_range[5]))

    def __getitem__(self, item):
        # base_points.py:276-346 — int / slice / mask / index array on the first axis, optional column
        # selection on the second
        cls = type(self)
        if isinstance(item, int):
            return cls(self.tensor[item].view(1, -1), points_dim=self.points_dim, attribute_dims=self.attribute_dims)
        if isinstance(item, np.ndarray):
            item = torch.from_numpy(item)
        t = self.tensor[item]
        assert t.dim() == 2, f'Indexing on Points with {item} failed to return a matrix!'
        return cls(t, points_dim=t.shape[1], attribute_dims=self.attribute_dims if t.shape[1] == self.points_dim else None)

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return self.__class__.__name__ + '(\n    ' + str(self.tensor) + ')'

    @classmethod
    def cat(cls, points_list):
        assert isinstance(points_list, (list, tuple))
        if len(points_list) == 0:
            return cls(torch.empty(0))
        assert all(isinstance(p, cls) for p in points_list)
        return cls(torch.cat([p.tensor for p in points_list], dim=0), points_dim=points_list[0].tensor.shape[1],
                   attribute_dims=points_list[0].attribute_dims)

    def to(self, device):
        return type(self)(self.tensor.to(device), points_dim=self.points_dim, attribute_dims=self.attribute_dims)

    def clone(self):
        return type(self)(self.tensor.clone(), points_dim=self.points_dim, attribute_dims=self.attribute_dims)

    def new_point(self, data):
        t = self.tensor.new_tensor(data) if not isinstance(data, torch.Tensor) else data.to(self.device)
        return type(self)(t, points_dim=self.points_dim, attribute_dims=self.attribute_dims)


class LiDARPoints(BasePoints):
    def __init__(self, tensor, points_dim=3, attribute_dims=None):
        super().__init__(tensor, points_dim=points_dim, attribute_dims=attribute_dims)
        self.rotation_axis = 2
