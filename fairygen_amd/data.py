"""Output sink: ``save_video(frames, path, fps, quality)`` (diffsynth/utils/data/__init__.py:140-145).

imageio/ffmpeg are not installed in this image; with them present the reference's writer call is used,
otherwise frames are written losslessly as an ``.npz`` next to the requested path (said in the return value).
"""
import numpy as np


def save_video(frames, save_path, fps, quality=9, ffmpeg_params=None):
    try:
        import imageio
    except ModuleNotFoundError:
        out = save_path + ".npz"
        np.savez_compressed(out, frames=np.stack([np.array(f) for f in frames]), fps=fps)
        return out
    writer = imageio.get_writer(save_path, fps=fps, quality=quality, ffmpeg_params=ffmpeg_params)
    for frame in frames:
        writer.append_data(np.array(frame))
    writer.close()
    return save_path
