"""KITTI calibration and object labels (host side of SURVEY 8f-4).

Same classes, attribute names and frame conventions as the reference's ``utils/data_classes.py``
(KITTICalibration :10-112, KITTISingleObject :117-282, KITTIMultiObjects :286-385); the hot path itself only reads
``calib.P2``.  Built differently: an object keeps its label-file state (camera 0, bottom-centre, global yaw) and the
three conversions are a view on it, so ``projected_center`` / ``projected_kpts`` never have to convert the state back
and forth.  Pinned against the reference classes by tests/golden/kitti_objects.npz (tests/test_input_pipeline.py).
"""
from typing import Any, Dict, List, Union

import numpy as np

from utils.geometry_ops import center_to_corner_box3d, points_cam2img, view_points

_CALIB_KEYS = ('P0', 'P1', 'P2', 'P3', 'R0', 'Tr_velo2cam', 'Tr_imu2velo')
_CALIB_SHAPES = {'R0': (3, 3)}
CLASS_TO_INDEX = {'DontCare': -1, 'Pedestrian': 0, 'Cyclist': 1, 'Car': 2}


def _rigid_inverse(tr: np.ndarray) -> np.ndarray:
    inv = np.zeros_like(tr)
    rt = tr[:3, :3].T
    inv[:3, :3] = rt
    inv[:3, 3] = -rt @ tr[:3, 3]
    return inv


class KITTICalibration:
    """``calib`` is a KITTI calib/*.txt path (first seven lines: P0..P3, R0_rect, Tr_velo_to_cam, Tr_imu_to_velo) or
    a dict of the matrices under the keys P0, P1, P2, P3, R0, Tr_velo2cam, Tr_imu2velo."""

    def __init__(self, calib: Union[Dict[str, Any], str]):
        if isinstance(calib, str):
            calib = self._from_file(calib)
        get = calib.get if hasattr(calib, 'get') else (lambda k, d=None: calib[k])
        self.P0, self.P1, self.P2, self.P3 = (np.asarray(calib[k], np.float32).reshape(3, 4) for k in ('P0', 'P1', 'P2', 'P3'))
        self.R0 = np.asarray(get('R0', np.eye(3)), np.float32).reshape(3, 3)
        self.V2C = np.asarray(get('Tr_velo2cam', np.eye(4)[:3]), np.float32).reshape(3, 4)
        self.I2V = np.asarray(get('Tr_imu2velo', np.eye(4)[:3]), np.float32).reshape(3, 4)
        self.C2V, self.V2I = _rigid_inverse(self.V2C), _rigid_inverse(self.I2V)
        self._refresh_intrinsics()

    def _refresh_intrinsics(self):
        self.cu, self.cv, self.fu, self.fv = self.P2[0, 2], self.P2[1, 2], self.P2[0, 0], self.P2[1, 1]
        self.tx, self.ty = self.P2[0, 3] / (-self.fu), self.P2[1, 3] / (-self.fv)

    @staticmethod
    def _from_file(path: str) -> Dict[str, np.ndarray]:
        out = {}
        with open(path) as f:
            for key, line in zip(_CALIB_KEYS, f.readlines()):
                out[key] = np.array(line.strip().split(' ')[1:], dtype=np.float32).reshape(_CALIB_SHAPES.get(key, (3, 4)))
        missing = [k for k in _CALIB_KEYS if k not in out]
        if missing:
            raise ValueError("%s: calibration lines missing for %s" % (path, missing))
        return out

    def get_info_dict(self) -> Dict[str, np.ndarray]:
        def pad(m):
            v = np.eye(4)
            v[:m.shape[0], :m.shape[1]] = m
            return v
        return {'P0': pad(self.P0), 'P1': pad(self.P1), 'P2': pad(self.P2), 'P3': pad(self.P3), 'R0_rect': pad(self.R0),
                'Tr_velo_to_cam': pad(self.V2C), 'Tr_imu_to_velo': pad(self.I2V)}

    def inverse_rigid_trans(self, tr):
        return _rigid_inverse(np.asarray(tr))

    def rescale(self, scale_x: float = None, scale_y: float = None) -> None:
        sx, sy = (1.0 if scale_x is None else scale_x), (1.0 if scale_y is None else scale_y)
        for m in (self.P0, self.P1, self.P2, self.P3):
            m[0, [0, 2, 3]] *= sx
            m[1, [1, 2, 3]] *= sy
        self._refresh_intrinsics()


class KITTISingleObject:
    """One line of a KITTI label_2 file.  ``loc`` / ``ry`` are reported in the CURRENT frame (``base_cam`` 0 or 2,
    ``yaw_type`` 'global' or 'local', ``center_type`` 'bottom-center' or 'gravity-center'); the label-file state
    itself is never modified."""

    def __init__(self, parsed_line: str, calib: KITTICalibration):
        self.calib, self.parsed_line = calib, parsed_line
        f = parsed_line.strip().split(' ')
        self.cls_str = f[0]
        self.cls_num = CLASS_TO_INDEX.get(self.cls_str, -1)
        self.truncation, self.occlusion, self.alpha = float(f[1]), float(f[2]), float(f[3])
        self.box2d = np.array([float(v) for v in f[4:8]], dtype=np.float32)
        self.h, self.w, self.l = float(f[8]), float(f[9]), float(f[10])
        self.dim = np.array((self.l, self.h, self.w), dtype=np.float32)
        self._loc0 = np.array([float(v) for v in f[11:14]], dtype=np.float32)      # camera 0, bottom centre
        self._ry_global = float(f[14])
        self.score = float(f[15]) if len(f) == 16 else -1.0
        self.dis_to_cam = np.linalg.norm(self._loc0)
        self.base_cam, self.yaw_type, self.center_type = 0, 'global', 'bottom-center'
        self._shift = np.zeros(3, dtype=np.float32)                                # translate() / flip() by augmentations
        self._flipped = False
        self.level_str = None
        self.level = self.get_obj_level()

    # ---- difficulty (KITTI benchmark definition)
    def get_obj_level(self) -> int:
        height = float(self.box2d[3]) - float(self.box2d[1]) + 1
        if self.truncation == -1:
            self.level_str = 'DontCare'
            return 0
        for level, name, (hmin, tmax, omax) in ((1, 'Easy', (40, 0.15, 0)), (2, 'Moderate', (25, 0.3, 1)), (3, 'Hard', (25, 0.5, 2))):
            if height >= hmin and self.truncation <= tmax and self.occlusion <= omax:
                self.level_str = name
                return level
        self.level_str = 'UnKnown'
        return 4

    # ---- frames
    def _cam_offset(self, cam: int) -> float:
        """x offset of camera ``cam`` relative to camera 0 (baseline / focal length)"""
        dst, src = getattr(self.calib, 'P%d' % cam), self.calib.P0
        return (dst[0, 3] - src[0, 3]) / dst[0, 0]

    def _location(self, cam: int, center: str) -> np.ndarray:
        loc = self._loc0.copy()
        if self._flipped:
            loc *= np.array((-1, 1, 1), dtype=np.float32)
        loc += self._shift
        if cam != 0:
            loc += np.array([self._cam_offset(cam), 0.0, 0.0])
        if center == 'gravity-center':
            loc += np.array([0.0, -0.5 * self.h, 0.0])
        return loc

    @property
    def loc(self) -> np.ndarray:
        return self._location(self.base_cam, self.center_type)

    @property
    def ry(self) -> float:
        if self.yaw_type == 'global':
            return self._ry_global
        x, _, z = self.loc
        return self._ry_global - np.arctan2(x, z)

    def translate(self, shift_x: float, shift_y: float = 0.0, shift_z: float = 0.0) -> None:
        self._shift += np.array([shift_x, shift_y, shift_z], dtype=np.float32)

    def flip(self) -> None:
        self._flipped = not self._flipped
        self._shift *= np.array((-1, 1, 1), dtype=np.float32)

    def convert_yaw(self, src_type: str, dst_type: str) -> None:
        self.yaw_type = dst_type

    def convert_cam(self, src_cam: int, dst_cam: int) -> None:
        self.base_cam = dst_cam

    def convert_center(self, src_type: str, dst_type: str) -> None:
        self.center_type = dst_type

    # ---- projections (always formed from the camera-0 / gravity-centre / global-yaw view, through P2)
    @property
    def projected_center(self) -> np.ndarray:
        return points_cam2img(self._location(0, 'gravity-center')[None], self.calib.P2, with_depth=True)[0]

    @property
    def projected_kpts(self):
        """(9,3): the eight box corners + the centre in pixels; column 2 = 1 where the corner lies in front of the
        camera (the dataset raises it to 2 for keypoints inside the image).  None for an object behind the camera."""
        center = self.projected_center
        if center[-1] <= 0:
            return None
        loc = self._location(0, 'gravity-center')
        corners = center_to_corner_box3d(loc[None], self.dim[None], np.array([self._ry_global]), origin=(0.5, 0.5, 0.5), axis=1)[0].T
        uv = view_points(corners, self.calib.P2, normalize=True).T[:, :2]
        pts = np.concatenate([np.hstack([uv, (corners[2] > 0).astype(np.float64)[:, None]]), [[center[0], center[1], 1.0]]])
        return pts

    @property
    def is_ignored(self) -> bool:
        return self.cls_num == -1


class KITTIMultiObjects:
    def __init__(self, obj_list: List[KITTISingleObject], ignore_dontcare: bool = True):
        self.ignore_dontcare = ignore_dontcare
        self.ori_obj_list = obj_list
        self.obj_list = [o for o in obj_list if not o.is_ignored] if ignore_dontcare else list(obj_list)
        self.calib = self.obj_list[0].calib if self.obj_list else None

    def __len__(self) -> int:
        return len(self.obj_list)

    def __getitem__(self, idx: int) -> KITTISingleObject:
        return self.obj_list[idx]

    def __repr__(self) -> str:
        return "KITTIMultiObjects(Objects: %d)" % len(self)

    def convert_yaw(self, src_type: str, dst_type: str) -> None:
        for o in self.obj_list:
            o.convert_yaw(src_type, dst_type)

    def convert_cam(self, src_cam: int, dst_cam: int) -> None:
        for o in self.obj_list:
            o.convert_cam(src_cam, dst_cam)

    def convert_center(self, src_type: str, dst_type: str) -> None:
        for o in self.obj_list:
            o.convert_center(src_type, dst_type)

    @property
    def original_objects(self):
        return KITTIMultiObjects(self.ori_obj_list, ignore_dontcare=False) if self.ignore_dontcare else self

    @property
    def info_dict(self) -> Dict[str, np.ndarray]:
        """the annotation dict of the KITTI evaluator (name, truncated, occluded, alpha, bbox, dimensions, location,
        rotation_y, score), one row per object"""
        cols = {'name': 'cls_str', 'truncated': 'truncation', 'occluded': 'occlusion', 'alpha': 'alpha', 'bbox': 'box2d',
                'dimensions': 'dim', 'location': 'loc', 'rotation_y': 'ry', 'score': 'score'}
        out = {}
        for key, attr in cols.items():
            vals = [getattr(o, attr) for o in self.obj_list]
            out[key] = np.stack(vals) if vals and isinstance(vals[0], np.ndarray) else np.array(vals)
        return out

    @staticmethod
    def get_objects_from_label(label_file: str, calibration: KITTICalibration):
        with open(label_file, 'r') as f:
            lines = [ln for ln in f.readlines() if ln.strip()]
        return KITTIMultiObjects([KITTISingleObject(ln, calibration) for ln in lines])
