"""`FLMRModelForRetrieval` -- the late-interaction SURFACE of the FLMR / ColBERT retriever models, over injected encoders.

north_star names the class `FLMRModelForRetrieval`; it is defined in the separate LinWeizheDragon/FLMR repository, which is
not part of the reference checkout, so parity with THAT class is UNPINNED.  What is mirrored here are its in-repo equivalents
(SURVEY 8b): `FLMR.query(input_ids, attention_mask, image_features)` (src/models/retriever/FLMR.py:73-99) and the methods
the four FLMR model classes inherit from ColBERT -- `doc(input_ids, attention_mask, keep_dims)` (TPC/modeling/colbert.py:
194-215), `score(Q, D_padded, D_mask)` (:217-224), `mask(input_ids, skiplist)` (:226-228), `forward(Q, D)` (:64-80).

Scope: the BERT / ViT / projection forward passes are outside this build -- they are injected:
    text_encoder(input_ids, attention_mask)  -> token states already projected to `dim` ([B, L, dim]; `linear(bert(...)[0])`)
    vision_projection(image_features)        -> [B, n_visual_tokens * dim]   (the mapping network; optional)
Everything after them is this class: the punctuation / padding mask, the concatenation of the projected visual tokens behind
the text tokens, the L2 normalisation, `keep_dims` handling, and the MaxSim score, which runs on the HIP padded scorer
(`flmr_colbert_score_padded`) -- forward only: `score` raises under autograd instead of silently computing without a graph.
`forward` returns the scores of every query against its `nway` passages; the in-batch-negative LOSS of training
(`compute_ib_loss_new`, colbert.py:82-128) is not part of the retrieval path and raises if requested.
"""
import string

import torch

from . import scoring


class FLMRModelForRetrieval:
    def __init__(self, text_encoder, doc_encoder=None, vision_projection=None, colbert_config=None, skiplist=None,
                 mask_punctuation_ids=None, dim=128, device=None):
        self.text_encoder = text_encoder
        self.doc_encoder = doc_encoder or text_encoder
        self.vision_projection = vision_projection
        self.colbert_config = colbert_config
        self.lm_embedding_size = dim
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        # colbert.py:24-28: with mask_punctuation the skiplist holds the punctuation symbols AND their token ids
        self.skiplist = dict(skiplist) if skiplist is not None else {}
        if mask_punctuation_ids is not None:
            self.skiplist.update({w: True for w in string.punctuation})
            self.skiplist.update({int(t): True for t in mask_punctuation_ids})
        self.use_gpu = False   # the reference halves D on its CUDA path (colbert.py:206-207); this build keeps fp32 numerics

    # ---- colbert.py:226-228 ------------------------------------------------------------------------------------------
    def mask(self, input_ids, skiplist):
        return [[(x not in skiplist) and (x != 0) for x in d] for d in input_ids.cpu().tolist()]

    # ---- FLMR.py:73-99 ------------------------------------------------------------------------------------------------
    def query(self, input_ids, attention_mask, image_features=None):
        input_ids, attention_mask = input_ids.to(self.device), attention_mask.to(self.device)
        Q = self.text_encoder(input_ids, attention_mask)
        mask = torch.tensor(self.mask(input_ids, skiplist=[]), device=self.device).unsqueeze(2).float()
        Q = Q * mask
        if image_features is not None:
            if self.vision_projection is None:
                raise ValueError("image_features given but no vision_projection was injected")
            last_hidden_states = self.vision_projection(image_features.to(self.device))
            last_hidden_states = last_hidden_states.reshape(last_hidden_states.shape[0], -1, self.lm_embedding_size)
            Q = torch.cat([Q, last_hidden_states], dim=1)
        return torch.nn.functional.normalize(Q, p=2, dim=2)

    # ---- colbert.py:194-215 -------------------------------------------------------------------------------------------
    def doc(self, input_ids, attention_mask, keep_dims=True):
        assert keep_dims in [True, False, "return_mask"]
        input_ids, attention_mask = input_ids.to(self.device), attention_mask.to(self.device)
        D = self.doc_encoder(input_ids, attention_mask)
        mask = torch.tensor(self.mask(input_ids, skiplist=self.skiplist), device=self.device).unsqueeze(2).float()
        D = D * mask
        D = torch.nn.functional.normalize(D, p=2, dim=2)
        if keep_dims is False:
            D, mask = D.cpu(), mask.bool().cpu().squeeze(-1)
            return [d[mask[idx]] for idx, d in enumerate(D)]
        if keep_dims == "return_mask":
            return D, mask.bool()
        return D

    # ---- colbert.py:217-224 -------------------------------------------------------------------------------------------
    def score(self, Q, D_padded, D_mask):
        if self.colbert_config is not None and getattr(self.colbert_config, "similarity", "cosine") == "l2":
            # colbert.py:220-222: the squared-distance variant is a plain torch expression upstream (no mask, no kernel): kept as it is
            assert getattr(self.colbert_config, "interaction", "colbert") == "colbert"
            return (-1.0 * ((Q.unsqueeze(2) - D_padded.unsqueeze(1)) ** 2).sum(-1)).max(-1).values.sum(-1)
        return scoring.colbert_score(Q, D_padded, D_mask, config=self.colbert_config)

    # ---- colbert.py:64-80 ---------------------------------------------------------------------------------------------
    def forward(self, Q, D):
        Q = self.query(*Q)
        D, D_mask = self.doc(*D, keep_dims="return_mask")
        nway = getattr(self.colbert_config, "nway", None) or (D.size(0) // Q.size(0))
        Q_duplicated = Q.repeat_interleave(nway, dim=0).contiguous()
        scores = self.score(Q_duplicated, D, D_mask)
        if self.colbert_config is not None and getattr(self.colbert_config, "use_ib_negatives", False) and torch.is_grad_enabled():
            raise NotImplementedError("the in-batch-negative loss (colbert.py:82-128) belongs to training; call forward under "
                                      "torch.no_grad() for the retrieval scores")
        return scores

    __call__ = forward
