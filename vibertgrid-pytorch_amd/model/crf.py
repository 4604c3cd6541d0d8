"""Linear-chain CRF layer of the `crf` classifier (reference model/crf.py:33-157) on libvbg kernels.

Same parameter (`transitions[i, j]` = score of the transition j -> i, with the START row / STOP column pinned to -10000) and the same
per-document API (`forward(feats, tags)` -> (log Z - gold score) / len, `inference(feats)` -> (path score, tag list)); `nll` /
`decode` take all documents of a batch at once (row ranges `doc_off`), one launch each: forward algorithm + gold score,
backward = (marginals - gold indicators), Viterbi with first-maximum tie breaking like `torch.max`.
"""
import torch
import torch.nn as nn

from vbg import functions as Fn
from vbg import ops

START_TAG = "<START>"
STOP_TAG = "<STOP>"


class CRF(nn.Module):
    def __init__(self, tag_to_ix):
        super().__init__()
        self.tag_to_ix = tag_to_ix
        self.tagset_size = len(tag_to_ix)
        assert self.tagset_size <= 64, "the CRF kernels hold one tag per lane of a wave (<= 64 tags)"
        self.transitions = nn.Parameter(torch.randn(self.tagset_size, self.tagset_size))
        self.transitions.data[tag_to_ix[START_TAG], :] = -10000
        self.transitions.data[:, tag_to_ix[STOP_TAG]] = -10000

    def _ends(self):
        return self.tag_to_ix[START_TAG], self.tag_to_ix[STOP_TAG]

    def nll(self, feats: torch.Tensor, tags_i32: torch.Tensor, doc_off: torch.Tensor) -> torch.Tensor:
        """feats [N, T], tags int32 [N], doc_off int32 [ndoc + 1] -> per-document (log Z - gold) / len, shape [ndoc]"""
        s, e = self._ends()
        return Fn.CrfNllFn.apply(feats, self.transitions, tags_i32.contiguous(), doc_off, s, e)

    def decode(self, feats: torch.Tensor, doc_off: torch.Tensor):
        """-> (best tag per row int32 [N], path score per document [ndoc])"""
        s, e = self._ends()
        return ops.crf_viterbi(feats.contiguous(), doc_off, self.transitions.detach().contiguous(), s, e)

    def forward(self, feats, tags):
        off = torch.tensor([0, feats.shape[0]], dtype=torch.int32).to(feats.device)
        return self.nll(feats, tags.int(), off)

    def inference(self, feats):
        off = torch.tensor([0, feats.shape[0]], dtype=torch.int32).to(feats.device)
        path, score = self.decode(feats.detach(), off)
        return score[0], path.tolist()
