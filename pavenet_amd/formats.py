"""Wire / on-disk formats either side of the path.

* ``bbox_kpt2result`` lives in detectors.py (opera/core/keypoint/transforms.py:132-154).
* ``kpt2json`` / ``results2json``: COCO-style keypoint json of a list of per-image results,
  restated from opera/datasets/posetrack_video_pose.py:268-348 (the dataset methods
  ``_kpt2json`` / ``results2json``; only the keypoint branch is on this path).
* ``load_checkpoint``: reads a reference checkpoint (``{'state_dict': ...}`` or a bare state
  dict, optional ``module.`` prefix from DDP, tools/test.py:226) into a model built here --
  the parameter names are the reference's own (tests/golden/state_dict_keys.json).
"""
import json

import numpy as np
import torch


def xyxy2xywh(bbox):
    """CocoDataset.xyxy2xywh."""
    _bbox = np.asarray(bbox).tolist()
    return [_bbox[0], _bbox[1], _bbox[2] - _bbox[0], _bbox[3] - _bbox[1]]


def kpt2json(results, img_ids, cat_ids=(1,)):
    """results[idx] = (det, kpt) with det[label] [n, 5], kpt[label] [n, K, 3] (numpy), as
    VideoPoseV1.simple_test returns them.  -> (bbox_json_results, kpt_json_results)."""
    bbox_json_results, kpt_json_results = [], []
    for idx, img_id in enumerate(img_ids):
        det, kpt = results[idx]
        for label in range(len(det)):
            bboxes = det[label]
            for i in range(bboxes.shape[0]):
                bbox_json_results.append(dict(image_id=img_id, bbox=xyxy2xywh(bboxes[i]),
                                              score=float(bboxes[i][4]),
                                              category_id=cat_ids[label]))
            kpts = kpt[label]
            for i in range(bboxes.shape[0]):
                kpt_json_results.append(dict(image_id=img_id, score=float(bboxes[i][4]),
                                             category_id=cat_ids[label],
                                             keypoints=kpts[i].reshape(-1).tolist()))
    return bbox_json_results, kpt_json_results


def results2json(results, img_ids, outfile_prefix, cat_ids=(1,)):
    """Writes ``<prefix>.keypoints.json`` (the reference dumps only the keypoint list for
    keypoint results, posetrack_video_pose.py:330-337) and returns the file map."""
    if not (isinstance(results[0], tuple) and isinstance(results[0][-1][0], np.ndarray)
            and results[0][-1][0].ndim == 3):
        raise TypeError('invalid type of results (expected (bbox lists, keypoint lists))')
    _, kpt_json = kpt2json(results, img_ids, cat_ids)
    files = dict(bbox=f'{outfile_prefix}.bbox.json', proposal=f'{outfile_prefix}.bbox.json',
                 keypoints=f'{outfile_prefix}.keypoints.json')
    with open(files['keypoints'], 'w') as f:
        json.dump(kpt_json, f)
    return files


def load_checkpoint(model, filename, map_location='cpu', strict=False):
    """mmcv.runner.load_checkpoint subset: returns the checkpoint dict; reports (does not hide)
    missing / unexpected keys."""
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    sd = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd, strict=strict)
    if missing or unexpected:
        print(f'load_checkpoint: missing keys {list(missing)[:8]}{"..." if len(missing) > 8 else ""}'
              f', unexpected keys {list(unexpected)[:8]}{"..." if len(unexpected) > 8 else ""}')
    return ckpt
