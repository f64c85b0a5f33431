"""Calibration holder: the hot path only reads ``.P2`` (reference utils/data_classes.py:10-40;
model/dense_heads/monocon_heads.py:501,543).  Label parsing / geometry stay out of scope."""
import numpy as np


class KITTICalibration:
    def __init__(self, calib):
        if isinstance(calib, str):
            calib = self._from_file(calib)
        self.P0, self.P1, self.P2, self.P3 = (np.asarray(calib[k], np.float32).reshape(3, 4) for k in ('P0', 'P1', 'P2', 'P3'))
        self.R0 = np.asarray(calib.get('R0', np.eye(3)), np.float32).reshape(3, 3)
        self.V2C = np.asarray(calib.get('Tr_velo2cam', np.eye(4)[:3]), np.float32).reshape(3, 4)
        self.cu, self.cv, self.fu, self.fv = self.P2[0, 2], self.P2[1, 2], self.P2[0, 0], self.P2[1, 1]
        self.tx, self.ty = self.P2[0, 3] / (-self.fu), self.P2[1, 3] / (-self.fv)

    @staticmethod
    def _from_file(path):
        keys = ('P0', 'P1', 'P2', 'P3', 'R0', 'Tr_velo2cam', 'Tr_imu2velo')
        out = {}
        with open(path) as f:
            for k, line in zip(keys, f.readlines()):
                out[k] = np.array(line.strip().split(' ')[1:], dtype=np.float32)
        return out
