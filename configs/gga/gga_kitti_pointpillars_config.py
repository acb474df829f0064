# GGA head on the PointPillars trunk (BASELINE.json config #2). The reference ships the
# pieces (PillarFeatureNet, PointPillarsScatter: configs/_base_/models/
# hv_pointpillars_secfpn_kitti.py:1-19) but no configs/gga/* file wires them to
# CenterHead_GGA; this file is that wiring.
_base_ = ['./gga_kitti_config.py']
voxel_size = [0.16, 0.16, 4]
point_cloud_range = [0, -39.68, -3, 69.12, 39.68, 1]
bn = dict(type='BN', eps=1e-3, momentum=0.01)

model = dict(
    pts_voxel_layer=dict(max_num_points=32, voxel_size=voxel_size, max_voxels=(16000, 40000),
                         point_cloud_range=point_cloud_range),
    pts_voxel_encoder=dict(_delete_=True, type='PillarFeatureNet', in_channels=4, feat_channels=[64],
                           with_distance=False, voxel_size=voxel_size, point_cloud_range=point_cloud_range),
    pts_middle_encoder=dict(_delete_=True, type='PointPillarsScatter', in_channels=64, output_shape=[496, 432]),
    pts_backbone=dict(_delete_=True, type='SECOND', in_channels=64, layer_nums=[3, 5, 5],
                      layer_strides=[2, 2, 2], out_channels=[64, 128, 256], norm_cfg=bn,
                      conv_cfg=dict(type='Conv2d', bias=False)),
    pts_neck=dict(_delete_=True, type='SECONDFPN', in_channels=[64, 128, 256], upsample_strides=[1, 2, 4],
                  out_channels=[128, 128, 128], norm_cfg=bn, upsample_cfg=dict(type='deconv', bias=False)),
    pts_bbox_head=dict(
        in_channels=384,
        bbox_coder=dict(post_center_range=point_cloud_range, out_size_factor=2, voxel_size=voxel_size[:2],
                        pc_range=point_cloud_range[:2])),
    train_cfg=dict(pts=dict(point_cloud_range=point_cloud_range, grid_size=[432, 496, 1], voxel_size=voxel_size,
                            out_size_factor=2)),
    test_cfg=dict(pts=dict(point_cloud_range=point_cloud_range, post_center_limit_range=point_cloud_range,
                           out_size_factor=2, voxel_size=voxel_size[:2])))
