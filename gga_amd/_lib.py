"""ctypes binding of ``libgga_hip.so`` (the C ABI declared in include/gga_hip.h).

The library is built in-tree by ``gga_amd/csrc/Makefile`` (``__graft_entry__.build()``)
and is the *only* compute path of the product: there is no CPU fallback. Loading
fails loudly when the shared object is missing, and every entry point raises
``RuntimeError`` with ``gga_last_error()`` on a non-zero status.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libgga_hip.so')
ABI_VERSION = 23

_lib = None

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
I3 = C.POINTER(C.c_int32 * 3)


class VoxelParams(C.Structure):
    _fields_ = [('voxel_size', C.c_float * 3), ('pc_range', C.c_float * 6),
                ('max_points', C.c_int32), ('max_voxels', C.c_int32)]


class PfnParams(C.Structure):
    _fields_ = [('voxel_size', C.c_float * 3), ('offsets', C.c_float * 3), ('eps', C.c_float),
                ('momentum', C.c_float), ('training', C.c_int32), ('in_features', C.c_int32),
                ('channels', C.c_int32)]


class LossParams(C.Structure):
    _fields_ = [('B', C.c_int32), ('K', C.c_int32), ('fm_w', C.c_int32),
                ('voxel_size', C.c_float * 2), ('out_size_factor', C.c_float),
                ('pc_range', C.c_float * 2), ('code_weights', C.c_float * 5),
                ('l1_loss_weight', C.c_float), ('w_bpl', C.c_float), ('w_srl', C.c_float),
                ('w_pal', C.c_float)]


# name -> (restype, argtypes); every symbol include/gga_hip.h declares
SIGNATURES = {
    'gga_last_error': (C.c_char_p, []),
    'gga_abi_version': (i32, []),
    'gga_timing_begin': (i32, [i32, i32, i64]),
    'gga_timing_collect': (i32, [i32, vp, i32]),
    'gga_voxel_grid_size': (None, [C.POINTER(VoxelParams), C.POINTER(C.c_int32 * 3)]),
    'gga_hard_voxelize_workspace_bytes': (sz, [i32, i64]),
    'gga_hard_voxelize_batch': (i32, [vp, i32, C.POINTER(C.c_int64), i32, C.POINTER(VoxelParams),
                                       vp, vp, vp, vp, vp, sz, vp]),
    'gga_hard_voxelize_prepared': (i32, [vp, i32, C.POINTER(C.c_int64), vp, i32, C.POINTER(VoxelParams),
                                          vp, vp, vp, vp, vp, sz, vp]),
    'gga_points_prepare_workspace_bytes': (sz, [i32, i64, i64]),
    'gga_points_prepare_batch': (i32, [vp, C.POINTER(C.c_int64), vp, C.POINTER(C.c_int64), vp, C.POINTER(C.c_int64),
                                        i32, i32, C.c_double, vp, C.POINTER(C.c_uint64), vp, vp, vp, sz, vp]),
    'gga_voxel_mean': (i32, [vp, vp, i64, i32, i32, i32, vp, vp]),
    'gga_pfn_workspace_bytes': (sz, [i64]),
    'gga_pfn_fwd': (i32, [vp, vp, vp, i64, vp, i32, C.POINTER(PfnParams), vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'gga_pfn_bwd': (i32, [vp, vp, vp, i64, vp, i32, C.POINTER(PfnParams), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'gga_pillar_scatter_map_bytes': (sz, [i32, i32, i32]),
    'gga_pillar_scatter_fwd': (i32, [vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    'gga_pillar_scatter_bwd': (i32, [vp, vp, i64, vp, i32, i32, i32, i32, i32, vp, vp]),
    'gga_pillar_conv_map': (i32, [vp, i64, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    'gga_sparse_index_bytes': (sz, [i64]),
    'gga_sparse_build_index': (i32, [vp, i64, i32, i32, i32, i32, vp, sz, vp]),
    'gga_sparse_out_index_bytes': (sz, [i64, i32]),
    'gga_sparse_out_sites_workspace_bytes': (sz, [i64, i32]),
    'gga_sparse_conv_out_sites': (i32, [vp, i64, i32, I3, I3, I3, I3, I3, vp, i64, vp, vp, sz, vp, sz, vp]),
    'gga_sparse_rulebook': (i32, [vp, i64, vp, i64, i32, I3, I3, I3, I3, I3, vp, i64, vp, i64, vp, vp, vp]),
    'gga_sparse_rowmask': (i32, [vp, i64, i32, vp, vp]),
    'gga_sparse_mask_order_workspace_bytes': (sz, [i64]),
    'gga_sparse_morton_order_workspace_bytes': (sz, [i64]),
    'gga_sparse_morton_order': (i32, [vp, i64, i32, i32, vp, vp, sz, vp]),
    'gga_sparse_mask_order': (i32, [vp, i64, i32, vp, vp, sz, vp]),
    'gga_sparse_packed_weight_bytes': (sz, [i32, i32, i32]),
    'gga_sparse_pack_weight': (i32, [vp, i32, i32, i32, i32, vp, vp]),
    'gga_sparse_split_weight_bytes': (sz, [i32, i32, i32]),
    'gga_sparse_pack_weight_split': (i32, [vp, i32, i32, i32, i32, vp, vp]),
    'gga_sparse_conv_apply_split': (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp]),
    'gga_sparse_conv_apply_split_strided': (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, vp]),
    'gga_sparse_conv_wgrad_split_strided': (i32, [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, vp, sz, vp]),
    'gga_sparse_conv_apply': (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp]),
    'gga_sparse_conv_wgrad': (i32, [vp, vp, vp, i64, i32, i32, i32, vp, vp]),
    'gga_sparse_conv_wgrad_workspace_bytes': (sz, [i64, i32, i32, i32]),
    'gga_sparse_conv_wgrad_split': (i32, [vp, vp, vp, i64, i32, i32, i32, vp, vp, sz, vp]),
    'gga_dense_conv3x3_pack': (i32, [vp, i64, i64, i64, i64, i32, i32, i32, vp, vp]),
    'gga_dense_wgrad3x3_workspace_bytes': (sz, [i32, i32, i32, i32, i32]),
    'gga_dense_wgrad3x3': (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i64, i64, i64, i64, i32, vp, sz, vp]),
    'gga_sparse_pack_weight_planes': (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    'gga_sparse_conv_apply_planes': (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, i32, vp, vp, vp]),
    'gga_sparse_conv_apply_stats': (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, i32, vp, vp, vp, vp]),
    'gga_sparse_conv_apply_tiles': (i64, [i64]),
    'gga_sparse_conv_apply_bn_bwd': (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp]),
    'gga_sparse_bev_nhwc_fwd': (i32, [vp, vp, i64, i32, i32, i32, i32, i32, vp, vp]),
    'gga_sparse_bev_nhwc_bwd': (i32, [vp, vp, i64, i32, i32, i32, i32, i32, vp, vp]),
    'gga_sparse_halo_tile_rows': (i64, []),
    'gga_sparse_halo_build': (i32, [vp, vp, i64, i64, i32, i32, vp, vp, vp, vp]),
    'gga_sparse_conv_apply_halo': (i32, [vp, vp, vp, vp, i32, vp, vp, i64, i64, i32, i32, i32, i32, vp, i64, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp]),
    'gga_sparse_conv_wgrad_planes': (i32, [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, i32, vp, vp, vp, sz, vp]),
    'gga_absmax_bits': (i32, [vp, i64, i32, i64, vp, vp]),
    'gga_dense_conv3x3_pack_planes': (i32, [vp, i64, i64, i64, i64, i32, i32, i32, i32, vp, vp, vp]),
    'gga_dense_conv3x3_planes': (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i64, i32, vp, i32, vp, vp, vp]),
    'gga_dense_conv3x3_levels': (i32, [i32, vp, vp, vp, vp, i32, i32, i32, vp, i64, i32, vp, vp, vp, i32, i32, vp, vp]),
    'gga_dense_conv3x3_bn_bwd_pays': (i32, [i32, i32, i32, i32]),
    'gga_dense_conv3x3_bn_bwd_pays_planes': (i32, [i32, i32, i32, i32, i32]),
    'gga_dense_conv3x3_bn_bwd': (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i64, i32, vp, i32, vp, vp, vp, i64, vp, vp, vp, vp, vp]),
    'gga_dense_wgrad3x3_planes': (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i64, i64, i64, i64, i32, i32, vp, vp, vp, sz, vp]),
    'gga_absmax_table_blocks': (i64, [i64]),
    'gga_absmax_table': (i32, [vp, i32, i64, vp, i32, vp]),
    'gga_pack_weights_table': (i32, [vp, i32, i64, i32, vp]),
    'gga_dense_wgrad3x3_block_amax': (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i64, i64, i64, i64, i32, i32, vp, i32, vp, i32, vp, sz, vp]),
    'gga_dense_conv3x3_tiles': (i64, [i32, i32, i32, i32]),
    'gga_dense_conv3x3_tiles_planes': (i64, [i32, i32, i32, i32, i32]),
    'gga_dense_conv3x3_tile_rows': (i32, [i32, i32, i32, i32, i32]),
    'gga_dense_conv3x3_stat_rows': (i64, [i32, i32, i32, i32, i32, i32]),
    'gga_dense_conv3x3_slice': (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i64, i32, vp, vp]),
    'gga_dense_conv3x3_stats': (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    'gga_bn_relu_fwd_partials': (i32, [vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, i32, vp, i64, vp, vp, vp, i32, vp, sz, vp]),
    'gga_bn_stats_partials': (i32, [vp, vp, vp, vp, i64, i32, f32, f32, vp, vp, vp, i32, vp]),
    'gga_bn_stats_partials_cols': (i32, [vp, vp, vp, vp, i64, i32, f32, f32, vp, vp, vp, i32, i32, i32, vp]),
    'gga_dense_conv3x3': (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    'gga_bn_relu_workspace_bytes': (sz, [i64, i32]),
    'gga_bn_relu_mask_bytes': (sz, [i64, i32]),
    'gga_bn_relu_fwd': (i32, [vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, i32, i32, vp, vp, vp, vp, sz, vp]),
    'gga_bn_relu_bwd': (i32, [vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    'gga_bn_relu_fwd_strided': (i32, [vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, i32, i32, vp, i64, vp, vp, vp, sz, vp]),
    'gga_bn_relu_bwd_strided': (i32, [vp, i64, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    'gga_bn_relu_fwd_ex': (i32, [vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, i32, i32, vp, i64, vp, vp, vp, i32, vp, vp, sz, vp]),
    'gga_bn_relu_bwd_ex': (i32, [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    'gga_column_sums': (i32, [vp, i64, i32, vp, vp, sz, vp]),
    'gga_bn_relu_bwd_partials': (i32, [vp, i64, vp, vp, vp, i64, i32, i32, vp, i32, vp, vp, vp, vp, vp, sz, vp]),
    'gga_gn_relu_workspace_bytes': (sz, [i32, i32]),
    'gga_gn_relu_fwd': (i32, [vp, vp, vp, i32, i64, i32, i32, f32, i32, vp, vp, vp, vp, vp, sz, vp]),
    'gga_gn_relu_bwd': (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    'gga_bn_stats': (i32, [vp, vp, vp, vp, vp, i64, i32, f32, f32, i32, vp, vp, vp, sz, vp]),
    'gga_head_conv3x3_fwd': (i32, [vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    'gga_head_conv3x3_workspace_bytes': (sz, [i32]),
    'gga_head_conv3x3_wgrad': (i32, [vp, i64, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    'gga_head_tail_bwd': (i32, [vp, vp, i64, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i64, vp, vp, vp, vp, sz, vp]),
    'gga_heatmap_splat': (i32, [vp, i32, i32, i32, vp, i32, vp, vp, i32, vp]),
    'gga_focal_loss_workspace_bytes': (sz, [i64]),
    'gga_focal_loss_fwd': (i32, [vp, vp, i64, f32, f32, f32, vp, vp, sz, vp]),
    'gga_focal_loss_bwd': (i32, [vp, vp, i64, f32, f32, f32, vp, vp, vp, vp]),
    'gga_gather_pred_fwd': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    'gga_gather_pred_bwd': (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    'gga_box_losses_workspace_bytes': (sz, [i32, i32]),
    'gga_box_losses_fwd': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, C.POINTER(LossParams),
                                  vp, vp, vp, vp, sz, vp]),
    'gga_box_losses_bwd': (i32, [vp, vp, i32, i32, vp, vp]),
    'gga_box_iou_rotated': (i32, [vp, i32, vp, i32, i32, i32, vp, vp]),
    'gga_nms_rotated_workspace_bytes': (sz, [i32]),
    'gga_nms_rotated_sorted': (i32, [vp, i32, f32, i32, vp, vp, vp, sz, vp]),
    'gga_circle_nms_sorted': (i32, [vp, i32, C.c_double, i32, vp, vp, vp, sz, vp]),
    'gga_region_grow_workspace_bytes': (sz, [i64, i32]),
    'gga_region_grow': (i32, [vp, i64, i32, vp, vp, vp, i32, C.c_double, i32, vp, vp, sz, vp]),
    'gga_points_in_convex_polyhedra': (i32, [vp, i64, i32, vp, vp, i32, i32, vp, vp]),
    'gga_plane_inliers': (i32, [vp, i64, i32, vp, i32, C.c_double, vp, vp, vp]),
    'gga_image_box_match': (i32, [vp, vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, vp]),
    'gga_centerpoint_detect_workspace_bytes': (sz, [i32, i32]),
    'gga_centerpoint_detect': (i32, [vp, vp, vp, i32, i32, i32, i32, vp, C.c_float, i32, C.c_float, vp, C.c_float, i32, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'gga_points_in_boxes': (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    'gga_fcos3d_targets': (i32, [vp, i32, i32, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float), f32, vp, i32,
                                  vp, vp, vp, vp, i32, vp, vp, vp, i64, i64, f32, vp, vp, vp, vp, vp, vp, vp]),
    'gga_dcn_im2col': (i32, [vp, vp, vp] + [i32] * 12 + [vp, vp]),
    'gga_dcn_im2col_amax': (i32, [vp, vp, vp] + [i32] * 12 + [vp, vp, vp]),
    'gga_dcn_col2im': (i32, [vp, vp, vp, vp] + [i32] * 12 + [vp, vp, vp, vp]),
}


TIME_SCATTER_FWD, TIME_DENSE_CONV, TIME_SPARSE_CONV, TIME_SPARSE_WGRAD, TIME_DENSE_WGRAD = range(5)


def timing_conv_key(cin, cout, hw=0):
    """GGA_TIMING_CONV_KEY of include/gga_hip.h."""
    return (int(cin) << 48) | (int(cout) << 32) | (int(hw) & 0xffffffff)


def timing_begin(site, max_samples, key=0):
    check(lib().gga_timing_begin(site, int(max_samples), int(key)), 'gga_timing_begin')


def timing_collect(site, cap):
    """-> list of per-call milliseconds of the armed session of `site` (waits for its events)."""
    buf = (C.c_float * max(int(cap), 1))()
    n = lib().gga_timing_collect(site, buf, int(cap))
    check(min(n, 0), 'gga_timing_collect')
    return [buf[i] for i in range(n)]


def build(force=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ['make', '-C', os.path.join(_HERE, 'csrc'), '-j4']
    if force:
        args.append('-B')
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: the GGA HIP kernels are not built. Run '
                '`python -c "import __graft_entry__ as g; g.build()"` (or `make -C gga_amd/csrc`). '
                'There is no CPU fallback for the product path.')
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if L.gga_abi_version() != ABI_VERSION:
            raise RuntimeError(f'libgga_hip.so ABI {L.gga_abi_version()} != binding ABI {ABI_VERSION}; rebuild')
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        raise RuntimeError(f'{what} failed (status {status}): {lib().gga_last_error().decode()}')
