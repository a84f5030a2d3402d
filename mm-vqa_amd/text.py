"""Text side of the data layer: WordPiece tokenisation, MLM keyword masking and the
[CLS] + visual slots + [SEP] + text + [SEP] + padding layout the hot path consumes.

Reference code mirrored here (bit-exact token / label / mask layout, same consumption of Python's `random` stream):
  get_keywords      pretrain/roco_utils.py:47-63   (incl. the `keywords.extend(word + '.')` quirk: it adds every
                                                    CHARACTER of every keyword, and '.', as keywords)
  mask_word         pretrain/roco_utils.py:135-160
  encode_text       pretrain/roco_utils.py:162-199 (task == 'MLM')
  encode_text_vqa   vqamed2019/utils.py:156-170    (hard-codes 5 visual slots and max_position_embeddings - 8)
  BertWordPiece     the `BertTokenizer` both scripts build (roco_utils.py:557, utils.py:222): BERT normaliser
                    (clean text, CJK spacing, accent stripping, lower-casing), whitespace + punctuation pre-tokeniser,
                    greedy longest-match WordPiece with '##' continuation pieces, [CLS] ... [SEP] template.
                    No network: the vocabulary comes from a local vocab.txt (one token per line).

This is host code (strings in, int64 tensors out); nothing here touches the GPU.  tests/test_text.py pins it against
fixtures produced by the REFERENCE's own functions (tests/golden/make_golden_text.py).
"""
from __future__ import annotations

import random as _random
import unicodedata

import torch


# --------------------------------------------------------------------------- tokenizer
def _is_whitespace(c):
    return c in "\t\n\r" or c.isspace() or unicodedata.category(c) == "Zs"


def _is_control(c):
    if c in "\t\n\r":
        return False
    return unicodedata.category(c) in ("Cc", "Cf", "Cn", "Co")


def _is_cjk(cp):
    return ((0x4E00 <= cp <= 0x9FFF) or (0x3400 <= cp <= 0x4DBF) or (0x20000 <= cp <= 0x2A6DF) or
            (0x2A700 <= cp <= 0x2B73F) or (0x2B740 <= cp <= 0x2B81F) or (0x2B920 <= cp <= 0x2CEAF) or
            (0xF900 <= cp <= 0xFAFF) or (0x2F800 <= cp <= 0x2FA1F))


def _is_punct(c):
    cp = ord(c)
    if (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126):
        return True
    return unicodedata.category(c).startswith("P")


class BertWordPiece:
    """uncased BERT tokenizer over a local vocabulary (dict token -> id, or a path to vocab.txt)"""

    SPECIALS = ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")

    def __init__(self, vocab, do_lower_case=True, max_input_chars_per_word=100):
        if isinstance(vocab, str):
            with open(vocab, "r", encoding="utf-8") as f:
                vocab = {tok.rstrip("\n"): i for i, tok in enumerate(f.readlines())}
        self.vocab = dict(vocab)
        self.do_lower_case = do_lower_case
        self.max_chars = max_input_chars_per_word
        self.unk_token, self.cls_token, self.sep_token, self.mask_token, self.pad_token = (
            "[UNK]", "[CLS]", "[SEP]", "[MASK]", "[PAD]")
        for t in self.SPECIALS:
            if t not in self.vocab:
                raise ValueError(f"vocabulary lacks {t}")
        self.unk_token_id = self.vocab["[UNK]"]
        self.cls_token_id = self.vocab["[CLS]"]
        self.sep_token_id = self.vocab["[SEP]"]
        self.mask_token_id = self.vocab["[MASK]"]
        self.pad_token_id = self.vocab["[PAD]"]

    # -- normaliser: clean -> CJK spacing -> strip accents -> lower
    def _normalize(self, text):
        out = []
        for c in text:
            cp = ord(c)
            if cp == 0 or cp == 0xFFFD or _is_control(c):
                continue
            out.append(" " if _is_whitespace(c) else c)
        text = "".join(out)
        out = []
        for c in text:
            if _is_cjk(ord(c)):
                out.extend((" ", c, " "))
            else:
                out.append(c)
        text = "".join(out)
        if self.do_lower_case:
            text = "".join(c for c in unicodedata.normalize("NFD", text) if unicodedata.category(c) != "Mn")
            text = text.lower()
        return text

    # -- pre-tokeniser: whitespace, then every punctuation character on its own
    @staticmethod
    def _pre_tokenize(text):
        words = []
        for chunk in text.split():
            cur = []
            for c in chunk:
                if _is_punct(c):
                    if cur:
                        words.append("".join(cur))
                        cur = []
                    words.append(c)
                else:
                    cur.append(c)
            if cur:
                words.append("".join(cur))
        return words

    def _wordpiece(self, word):
        if len(word) > self.max_chars:
            return [self.unk_token]
        pieces, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = word[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    cur = sub
                    break
                end -= 1
            if cur is None:
                return [self.unk_token]
            pieces.append(cur)
            start = end
        return pieces

    def _split_specials(self, text):
        """special tokens written out in the text are kept whole (they bypass normalisation)"""
        parts = [text]
        for sp in self.SPECIALS:
            nxt = []
            for p in parts:
                if isinstance(p, tuple):
                    nxt.append(p)
                    continue
                segs = p.split(sp)
                for i, s in enumerate(segs):
                    if i:
                        nxt.append((sp,))
                    if s:
                        nxt.append(s)
            parts = nxt
        return parts

    def tokenize(self, text):
        toks = []
        for part in self._split_specials(text):
            if isinstance(part, tuple):
                toks.append(part[0])
                continue
            for w in self._pre_tokenize(self._normalize(part)):
                toks.extend(self._wordpiece(w))
        return toks

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            return self.vocab.get(tokens, self.unk_token_id)
        return [self.vocab.get(t, self.unk_token_id) for t in tokens]

    def encode(self, text):
        """[CLS] + ids + [SEP]"""
        return [self.cls_token_id] + self.convert_tokens_to_ids(self.tokenize(text)) + [self.sep_token_id]


# --------------------------------------------------------------------------- ROCO / MLM (pretrain/roco_utils.py)
def get_keywords(med_vocab: dict):
    """roco_utils.py:47-63 on the unpickled med_vocab dict {category: [words]}.  Order of the returned list is the
    iteration order of a Python set (as in the reference); only membership is ever used (:141)."""
    keywords = []
    for _k, v in med_vocab.items():
        keywords.extend(v)
    keywords_ = list(set(keywords))
    for word in keywords_:
        keywords.extend(word + ".")       # extends by the CHARACTERS of word + '.'  (reference quirk, :59)
    return list(set(keywords))


def _masked_pieces(sentence, tokenizer, keywords, mlm_prob, rng):
    """Word pieces of a caption with keyword pieces masked: yields (piece or '[MASK]', label id) pairs.
    Contract with pretrain/roco_utils.py:135-160 (what the fixtures pin): words are the whitespace fields of the
    caption; only words that are themselves in the keyword set take part; EVERY piece of such a word consumes exactly
    one rng.random() draw, in reading order, and is masked when the draw is below mlm_prob; the label of a masked piece
    is the id of the first piece of that piece's own string re-tokenised (so '##ing' is labelled by the id of '#')."""
    kw = keywords if isinstance(keywords, (set, frozenset)) else set(keywords)
    for word in sentence.split():
        pieces = tokenizer.tokenize(word)
        if word not in kw:
            for pc in pieces:
                yield pc, 0
            continue
        hits = [rng.random() < mlm_prob for _ in pieces]
        for pc, hit in zip(pieces, hits):
            if hit:
                yield "[MASK]", tokenizer.encode(pc)[1]
            else:
                yield pc, 0


def mask_word(sentence, tokenizer, keywords, mlm_prob, rng=_random):
    """roco_utils.py:135-160 -> (word pieces with masks applied, label id per piece)"""
    pairs = list(_masked_pieces(sentence, tokenizer, keywords, mlm_prob, rng))
    return [pc for pc, _ in pairs], [lab for _, lab in pairs]


def _fill_row(ids, seg, mask, text_ids, n_vis, cls_id, sep_id):
    """One row of the model's token layout, written in place into zeroed length-T rows:
        position   0      1 .. n_vis     n_vis+1   n_vis+2 .. n_vis+1+n   n_vis+2+n   rest
        ids        [CLS]  0 (visual)     [SEP]     text piece ids         [SEP]       0
        seg        0      0              0         1                      1           0
        mask       1      1              1         1                      1           0
    (pretrain/roco_utils.py:176-186 and vqamed2019/utils.py:160-168 build the same rows by list concatenation.)
    Returns the position of the first text piece."""
    n, t0 = len(text_ids), n_vis + 2
    ids[0] = cls_id
    ids[n_vis + 1] = sep_id
    if n:
        ids[t0:t0 + n] = torch.as_tensor(text_ids, dtype=torch.long)
    ids[t0 + n] = sep_id
    seg[t0:t0 + n + 1] = 1
    mask[:t0 + n + 1] = 1
    return t0


def encode_text(caption, tokenizer, keywords, num_vis, max_position_embeddings, mlm_prob, rng=_random):
    """roco_utils.py:162-199 (task 'MLM') -> (tokens, segment_ids, input_mask, labels), int64 tensors of length
    max_position_embeddings; the caption keeps its first max_position_embeddings - (num_vis + 3) pieces (the RNG is
    still drawn for the pieces that are cut: masking happens before truncation)"""
    T = max_position_embeddings
    pieces, piece_labels = mask_word(caption, tokenizer, keywords, mlm_prob, rng)
    room = T - (num_vis + 3)
    text_ids = tokenizer.convert_tokens_to_ids(pieces)[:room]
    ids, seg, mask, lab = (torch.zeros(T, dtype=torch.long) for _ in range(4))
    t0 = _fill_row(ids, seg, mask, text_ids, num_vis, tokenizer.cls_token_id, tokenizer.sep_token_id)
    if text_ids:
        lab[t0:t0 + len(text_ids)] = torch.as_tensor(piece_labels[:room], dtype=torch.long)
    return ids, seg, mask, lab


# --------------------------------------------------------------------------- VQA-Med (vqamed2019/utils.py)
def encode_text_vqa(question, tokenizer, max_position_embeddings):
    """utils.py:156-170 -> (tokens, segment_ids, input_mask) as lists: always 5 visual slots, the question keeps its
    first max_position_embeddings - 8 pieces"""
    T = max_position_embeddings
    text_ids = tokenizer.convert_tokens_to_ids(tokenizer.tokenize(question))[:T - 8]
    ids, seg, mask = (torch.zeros(T, dtype=torch.long) for _ in range(3))
    _fill_row(ids, seg, mask, text_ids, 5, tokenizer.cls_token_id, tokenizer.sep_token_id)
    return ids.tolist(), seg.tolist(), mask.tolist()


# --------------------------------------------------------------------------- batches for the hot path
def roco_text_batch(captions, tokenizer, keywords, num_vis=5, max_position_embeddings=75, mlm_prob=0.15, rng=_random,
                    device=None):
    """what the ROCO DataLoader collates (roco_utils.py:573-587): (ids, segment_ids, mask, target), each [B, T]"""
    rows = [encode_text(c, tokenizer, keywords, num_vis, max_position_embeddings, mlm_prob, rng) for c in captions]
    out = tuple(torch.stack([r[i] for r in rows]) for i in range(4))
    return tuple(t.to(device) for t in out) if device is not None else out


def vqa_text_batch(questions, tokenizer, max_position_embeddings=28, device=None):
    """what the VQAMed DataLoader collates (utils.py:249-257): (ids, segment_ids, mask), each [B, T]"""
    rows = [encode_text_vqa(q, tokenizer, max_position_embeddings) for q in questions]
    out = tuple(torch.tensor([r[i] for r in rows], dtype=torch.long) for i in range(3))
    return tuple(t.to(device) for t in out) if device is not None else out
