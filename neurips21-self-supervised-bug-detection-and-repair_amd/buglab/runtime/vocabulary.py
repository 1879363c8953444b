"""`Vocabulary` and `split_identifier_into_parts` -- the two `dpu_utils` helpers the hot path's
collator side touches (reference buglab/models/basemodel.py:67-69,154;
buglab/representations/data.py:113,159).  Written from the call-site contract; dpu_utils itself is
not installable offline."""
import re
from collections import Counter
from typing import Dict, Iterable, List, Union

_CAMEL = re.compile(r"[A-Z]+(?=[A-Z][a-z])|[A-Z]?[a-z]+|[A-Z]+|[0-9]+|[^A-Za-z0-9]+")


def split_identifier_into_parts(identifier: str) -> List[str]:
    """snake_case and camelCase split, lower-cased; an identifier with no parts is returned whole."""
    parts: List[str] = []
    for piece in identifier.split("_"):
        if piece:
            parts.extend(m.group(0).lower() for m in _CAMEL.finditer(piece))
    return parts if parts else [identifier]


class Vocabulary:
    PAD, UNK = "%PAD%", "%UNK%"

    def __init__(self, add_unk: bool = True, add_pad: bool = False):
        self.token_to_id: Dict[str, int] = {}
        self.id_to_token: List[str] = []
        if add_pad:
            self.add_or_get_id(self.PAD)
        if add_unk:
            self.add_or_get_id(self.UNK)

    @staticmethod
    def get_pad() -> str:
        return Vocabulary.PAD

    @staticmethod
    def get_unk() -> str:
        return Vocabulary.UNK

    def add_or_get_id(self, token: str) -> int:
        i = self.token_to_id.get(token)
        if i is None:
            i = len(self.id_to_token)
            self.token_to_id[token] = i
            self.id_to_token.append(token)
        return i

    def get_id_or_unk(self, token: str) -> int:
        i = self.token_to_id.get(token)
        if i is not None:
            return i
        return self.token_to_id[self.UNK]  # KeyError if built with add_unk=False: same failure as dpu_utils

    def is_unk(self, token: str) -> bool:
        return token not in self.token_to_id

    def get_name_for_id(self, token_id: int) -> str:
        return self.id_to_token[token_id]

    def __len__(self) -> int:
        return len(self.id_to_token)

    @classmethod
    def create_vocabulary(cls, tokens: Union[Iterable[str], Counter], max_size: int, count_threshold: int = 5,
                          add_unk: bool = True, add_pad: bool = False) -> "Vocabulary":
        counts = tokens if isinstance(tokens, Counter) else Counter(tokens)
        vocab = cls(add_unk=add_unk, add_pad=add_pad)
        # deterministic: by descending count, then lexicographic
        for tok, c in sorted(counts.items(), key=lambda kv: (-kv[1], kv[0])):
            if len(vocab) >= max_size + (1 if add_unk else 0) + (1 if add_pad else 0):
                break
            if c >= count_threshold:
                vocab.add_or_get_id(tok)
        return vocab
