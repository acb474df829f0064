# Pseudo-label matching run (reference: configs/gga/gga_kitti_matching_config.py - gga_kitti_config.py with ONE line changed,
# :93 dataset_type = 'KittiDataset_GGA_match'): `tools/test.py <this config> <checkpoint> --eval mAP` runs the trained
# detector over kitti_infos_trainval_GGA.pkl and the dataset's `evaluate` matches the detections to the 2D boxes and writes
# the pseudo-label file (gga_amd/datasets.py::KittiDataset_GGA_match.evaluate -> pseudo_labels.pseudo_label_matching_kitti).
_base_ = './gga_kitti_config.py'
dataset_type = 'KittiDataset_GGA_match'
data_root = 'data/kitti/'
class_names = ['Pedestrian', 'Cyclist', 'Car']
point_cloud_range = [0, -40, -3, 70.4, 40, 1]
input_modality = dict(use_lidar=True, use_camera=True)
test_pipeline = [
    dict(type='LoadPointsFromFile', coord_type='LIDAR', load_dim=4, use_dim=4),
    dict(type='MultiScaleFlipAug3D', img_scale=(1333, 800), pts_scale_ratio=1, flip=False,
         transforms=[dict(type='GlobalRotScaleTrans', rot_range=[0, 0], scale_ratio_range=[1., 1.], translation_std=[0, 0, 0]),
                     dict(type='RandomFlip3D'),
                     dict(type='PointsRangeFilter', point_cloud_range=point_cloud_range),
                     dict(type='DefaultFormatBundle3D', class_names=class_names, with_label=False),
                     dict(type='Collect3D', keys=['points'])])]
_test = dict(type=dataset_type, data_root=data_root, ann_file=data_root + 'kitti_infos_trainval_GGA.pkl', split='training',
             pts_prefix='velodyne_reduced', pipeline=test_pipeline, modality=input_modality, classes=class_names, test_mode=True,
             box_type_3d='LiDAR')
data = dict(samples_per_gpu=32, workers_per_gpu=4, val=_test, test=_test)
