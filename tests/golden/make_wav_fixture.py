"""Packs the audio the reference's own tests hold into one tensor fixture (data, not source).

Run in the build container only (needs /root/reference):  python tests/golden/make_wav_fixture.py

/root/reference/tests/{data,new_data} hold 16 PCM WAV files -- 7 distinct recordings of LJ008-0217
(22 050 Hz, mono, int16, 7.58 s each): the LJSpeech original and its re-synthesis by six vocoders
(`new_data/A_ljspeech ... G_waveglow`; `data/` holds the same bytes under the older folder names).
This script stores the int16 samples of the 7 distinct files with their folder names.  Together
with the weights of the shipped `models/*packetssym5*.pt` checkpoint (already a tensor fixture,
tests/golden/dcnn_shipped_packetssym5.pt) they are the only reference-held numbers that DEPEND on
what ptwt computes: a classifier trained on ptwt's sym5 level-8 packets (frequency order,
reflect-padded analysis steps) must separate these recordings when it is fed ours.
"""

import hashlib
import os
import wave

import numpy as np
import torch

SRC = "/root/reference/tests/new_data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_wavs_lj008_0217.pt")


def main() -> None:
    names, pcm, digests = [], [], []
    for folder in sorted(os.listdir(SRC)):
        files = sorted(f for f in os.listdir(os.path.join(SRC, folder)) if f.endswith(".wav") and "copy" not in f)
        assert len(files) == 1, (folder, files)
        with wave.open(os.path.join(SRC, folder, files[0])) as w:
            assert (w.getframerate(), w.getnchannels(), w.getsampwidth()) == (22050, 1, 2)
            raw = w.readframes(w.getnframes())
        names.append(folder)
        pcm.append(torch.from_numpy(np.frombuffer(raw, dtype=np.int16).copy()))
        digests.append(hashlib.md5(raw).hexdigest())
    torch.save({"folders": names, "pcm_int16": pcm, "md5": digests, "sample_rate": 22050,
                "real_folder": "A_ljspeech", "trained_against": "B_fullbandmelgan"}, OUT)
    print(OUT, [(n, len(p)) for n, p in zip(names, pcm)])


if __name__ == "__main__":
    main()
