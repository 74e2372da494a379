import os, sys, torch, numpy as np
from clearconverse_amd.tokenizer import DecodeRules
from clearconverse_amd.weights import WhisperDims, synthetic_whisper_state_dict
from clearconverse_amd.whisper import WhisperModel
from clearconverse_amd.audio import synthetic_clip
B = int(sys.argv[1]); out = sys.argv[2]
dims = WhisperDims.small_en(); sd = synthetic_whisper_state_dict(dims, seed=0)
m = WhisperModel(dims, sd, max_batch=B)
rules = DecodeRules()
clip = synthetic_clip(0, 30.0)
dev = torch.from_numpy(np.stack([clip] * 32)).cuda()
for b0 in range(0, B, 32): pass
big = dev.repeat(B // 32, 1).contiguous()
m.log_mel(big, [len(clip)] * B); m.encode(B)
m.trace_lanes(out, 2)
for _ in range(2):
    m.decode_greedy([[rules.sot]] * B, sample_len=24)
torch.cuda.synchronize()
m.trace_lanes(None)
