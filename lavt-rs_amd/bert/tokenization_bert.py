"""`BertTokenizer` of the reference's `bert.tokenization_bert` (HF transformers 3.0.2, absent from the reference tree): uncased BERT
WordPiece, host side.  Call sites: data/dataset_refer_bert.py:55 (`BertTokenizer.from_pretrained(args.bert_tokenizer)`), :66
(`tokenizer.encode(text=sentence_raw, add_special_tokens=True)`), and the same in data/ytvos.py / a2d.py / davis.py.

The algorithm is the published one of BERT's `tokenization.py`: clean (drop control characters, map whitespace to a blank) -> put blanks
around CJK ideographs -> split on whitespace -> lower-case and strip accents (NFD, drop combining marks) -> split off every punctuation
character -> greedy longest-match-first WordPiece with the `##` continuation prefix, `[UNK]` for a word with no cover or over 100 chars.
`encode` wraps the ids in [CLS] ... [SEP].  `pad_ids` is the dataset's truncate / zero-pad / mask step (dataset_refer_bert.py:62-75).
"""
import os
import unicodedata


def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:        # ASCII non-alphanumerics count as punctuation ("$", "^", "`" ...)
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F
            or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class BertTokenizer:
    def __init__(self, vocab_file, do_lower_case=True, unk_token="[UNK]", cls_token="[CLS]", sep_token="[SEP]", pad_token="[PAD]",
                 max_input_chars_per_word=100):
        self.vocab = {}
        with open(vocab_file, encoding="utf-8") as f:
            for i, line in enumerate(f):
                self.vocab[line.rstrip("\n")] = i
        self.do_lower_case = do_lower_case
        self.unk_token, self.cls_token, self.sep_token, self.pad_token = unk_token, cls_token, sep_token, pad_token
        self.never_split = {unk_token, cls_token, sep_token, pad_token, "[MASK]"}
        self.max_chars = max_input_chars_per_word

    @classmethod
    def from_pretrained(cls, path, *unused, **kw):
        """`path`: a vocab.txt, or a directory holding one (what `args.bert_tokenizer` points at; there is no hub download)."""
        vf = os.path.join(str(path), "vocab.txt") if os.path.isdir(str(path)) else str(path)
        if not os.path.isfile(vf):
            raise OSError(f"BertTokenizer.from_pretrained: no vocab.txt at {path!r}")
        return cls(vf, **kw)

    # ---- basic tokenizer -----------------------------------------------------------------------------------------------
    def _basic(self, text):
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_whitespace(ch):
                out.append(" ")
            elif _is_cjk(cp):
                out.extend((" ", ch, " "))
            else:
                out.append(ch)
        words = []
        for tok in "".join(out).split():
            if tok in self.never_split:
                words.append(tok)
                continue
            if self.do_lower_case:
                tok = "".join(c for c in unicodedata.normalize("NFD", tok.lower()) if unicodedata.category(c) != "Mn")
            cur = []
            for ch in tok:
                if _is_punctuation(ch):
                    if cur:
                        words.append("".join(cur))
                        cur = []
                    words.append(ch)
                else:
                    cur.append(ch)
            if cur:
                words.append("".join(cur))
        return words

    # ---- WordPiece -----------------------------------------------------------------------------------------------------
    def _wordpiece(self, word):
        if len(word) > self.max_chars:
            return [self.unk_token]
        pieces, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = word[start:end] if start == 0 else "##" + word[start:end]
                if sub in self.vocab:
                    cur = sub
                    break
                end -= 1
            if cur is None:
                return [self.unk_token]
            pieces.append(cur)
            start = end
        return pieces

    def tokenize(self, text):
        toks = []
        for w in self._basic(text):
            toks.extend([w] if w in self.never_split else self._wordpiece(w))
        return toks

    def convert_tokens_to_ids(self, tokens):
        unk = self.vocab[self.unk_token]
        return [self.vocab.get(t, unk) for t in tokens]

    def encode(self, text, add_special_tokens=True):
        ids = self.convert_tokens_to_ids(self.tokenize(text))
        if add_special_tokens:
            ids = [self.vocab[self.cls_token]] + ids + [self.vocab[self.sep_token]]
        return ids


def pad_ids(token_ids, max_tokens):
    """data/dataset_refer_bert.py:62-75 -- truncate to max_tokens (20; 22 for the combined / video sets), zero-pad, attention mask on real ids"""
    token_ids = list(token_ids)[:max_tokens]
    ids, mask = [0] * max_tokens, [0] * max_tokens
    ids[:len(token_ids)] = token_ids
    mask[:len(token_ids)] = [1] * len(token_ids)
    return ids, mask
