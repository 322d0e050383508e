"""OthelloGame -- drop-in for Othello/__init__.py:1-275 whose rule evaluation runs on the HIP rule kernels.

Same names, arguments, return types and error behaviour as the reference class; the board is kept as the
reference's (n,n,2) bool array (ch0 BLACK, ch1 WHITE) and mutated IN PLACE by play() (R10/R11), every
legal-move / flip / terminal query is one call into libothellozero_amd (batch of one).
Bulk work should use the batched entry points (`rules_*` below, `training.selfplay_batch`).
"""
from enum import Enum, auto

import numpy as np

from . import _lib


class BoardView(Enum):           # member names and values are public surface (Othello/__init__.py:6-8): callers pass BoardView.TWO_CHANNELS
    ONE_CHANNEL = auto()
    TWO_CHANNELS = auto()


class OthelloPlayer(Enum):       # value = the disc's sign in the one-channel view (Othello/__init__.py:12-14)
    BLACK = 1
    WHITE = -1

    @property
    def opponent(self):
        return OthelloPlayer(-self.value)


# ---------------------------------------------------------------- batched rule calls (uint64 bitboards)
def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64).ravel()


def rules_legal_moves(own, opp, n):
    own, opp = _u64(own), _u64(opp)
    out = np.zeros(own.size, np.uint64)
    _lib.check(_lib.require_gpu().oz_rules_legal_moves(_lib.p_u64(own), _lib.p_u64(opp), n, own.size, _lib.p_u64(out)))
    return out


def rules_apply_moves(own, opp, sq, n):
    own, opp = _u64(own), _u64(opp)
    sq = np.ascontiguousarray(sq, dtype=np.uint8).ravel()
    o2, p2 = np.zeros(own.size, np.uint64), np.zeros(own.size, np.uint64)
    _lib.check(_lib.require_gpu().oz_rules_apply_moves(_lib.p_u64(own), _lib.p_u64(opp), _lib.p_u8(sq), n, own.size,
                                                       _lib.p_u64(o2), _lib.p_u64(p2)))
    return o2, p2


def rules_status(ch0, ch1, n):
    ch0, ch1 = _u64(ch0), _u64(ch1)
    k = ch0.size
    fin, p0, p1, win = np.zeros(k, np.uint8), np.zeros(k, np.int32), np.zeros(k, np.int32), np.zeros(k, np.int8)
    _lib.check(_lib.require_gpu().oz_rules_status(_lib.p_u64(ch0), _lib.p_u64(ch1), n, k, _lib.p_u8(fin), _lib.p_i32(p0),
                                                  _lib.p_i32(p1), _lib.p_i8(win)))
    return fin, p0, p1, win


def rules_play(black, white, player, sq, n):
    black, white = _u64(black), _u64(white)
    player = np.ascontiguousarray(player, dtype=np.int8).ravel()
    sq = np.ascontiguousarray(sq, dtype=np.uint8).ravel()
    k = black.size
    b2, w2, p2, f2 = np.zeros(k, np.uint64), np.zeros(k, np.uint64), np.zeros(k, np.int8), np.zeros(k, np.uint8)
    _lib.check(_lib.require_gpu().oz_rules_play(_lib.p_u64(black), _lib.p_u64(white), _lib.p_i8(player), _lib.p_u8(sq), n, k,
                                                _lib.p_u64(b2), _lib.p_u64(w2), _lib.p_i8(p2), _lib.p_u8(f2)))
    return b2, w2, p2, f2


def _mask_to_actions(mask):
    mask = int(mask)
    return [(s >> 3, s & 7) for s in range(64) if (mask >> s) & 1]      # ascending row-major (R4)


class OthelloGame:
    PLAYER_CHANNELS = {OthelloPlayer.BLACK: 0, OthelloPlayer.WHITE: 1}
    ALL_DIRECTIONS = np.array([(1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1)])

    def __init__(self, board_size=8, initial_board=None, current_player=OthelloPlayer.BLACK):
        # same checks and messages as Othello/__init__.py:105-108; the game keeps the caller's array (no copy, R11)
        if board_size % 2:
            raise AssertionError('Board size must be even')
        if initial_board is not None and initial_board.shape != (board_size, board_size, 2):
            raise AssertionError(f'Expecting initial board shape ({board_size}, {board_size}, 2)')
        assert board_size in (4, 6, 8), 'the HIP rule kernels cover board sizes 4, 6 and 8'
        given = initial_board is not None
        self._board = initial_board if given else OthelloGame.initial_board(board_size)
        self._n, self._round, self.current_player = board_size, 1, current_player
        self._flat_view = (None, None)              # (round it was built in, one-channel array)
        self._has_finished = given and OthelloGame.has_board_finished(self._board)

    @property
    def board_size(self):
        return self._n

    @property
    def round(self):
        return self._round

    def board(self, view=BoardView.ONE_CHANNEL):
        if view is BoardView.TWO_CHANNELS:
            return self._board                      # the live array, no copy (R11)
        if view is not BoardView.ONE_CHANNEL:
            raise TypeError('Expecting BoardView type')
        if self._flat_view[0] != self._round:       # rebuilt once per round, a fresh array each time (Othello:127-133)
            self._flat_view = (self._round, OthelloGame.convert_to_one_channel_board(self._board))
        return self._flat_view[1]

    def is_square_free(self, x, y):
        """Othello/__init__.py:88-98.  (The reference's body reads the undefined names `row` / `col` and raises NameError;
        the mirror answers the question its docstring asks.)"""
        return OthelloGame.is_board_square_free(self._board, x, y)

    def is_valid_action(self, row, col):
        return OthelloGame.is_valid_player_action(self._board, self.current_player, row, col)

    def get_valid_actions(self):
        return OthelloGame.get_player_valid_actions(self._board, self.current_player)

    def get_free_squares(self):
        return OthelloGame.get_board_free_squares(self._board)

    def has_finished(self):
        return self._has_finished

    def play(self, row, col):
        assert not self._has_finished, 'Game has ended'
        n = self._n
        black, white = _lib.pack_board(self._board)
        b2, w2, p2, f2 = rules_play([black], [white], [self.current_player.value], [int(row) * 8 + int(col)], n)
        self._board[...] = _lib.unpack_board(int(b2[0]), int(w2[0]), n)          # in place, like flip_board_squares
        self._round += 1
        self.current_player = OthelloPlayer(int(p2[0]))
        self._has_finished = bool(f2[0])

    def get_players_points(self):
        return OthelloGame.get_board_players_points(self._board)

    def get_winning_player(self):
        return OthelloGame.get_board_winning_player(self._board)

    # ---- static board functions (Othello/__init__.py:177-274)
    @staticmethod
    def initial_board(board_size):
        if board_size % 2:
            raise AssertionError('Board size must be even')
        b = np.zeros((board_size, board_size, 2), dtype=bool)
        lo, hi = board_size // 2 - 1, board_size // 2
        b[lo, hi, 0] = b[hi, lo, 0] = True          # BLACK on the anti-diagonal of the centre (Othello:177-184)
        b[lo, lo, 1] = b[hi, hi, 1] = True          # WHITE on the diagonal
        return b

    ALL_DIRECTIONS = np.array([(1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1)])    # Othello/__init__.py:27 (same order)

    @staticmethod
    def get_all_directions_squares(board_size, row, col):
        """the eight rays out of (row, col), in ALL_DIRECTIONS order, each as a generator of squares (Othello/__init__.py:186-189).
        Host-side geometry for callers of the static surface; the kernels walk rays as bit shifts."""
        return (OthelloGame.get_direction_squares(board_size, step, row, col) for step in OthelloGame.ALL_DIRECTIONS)

    @staticmethod
    def get_direction_squares(board_size, direction, row, col):
        """squares of the ray that leaves (row, col) along `direction`, nearest first, up to the edge (Othello/__init__.py:191-198)"""
        dr, dc = int(direction[0]), int(direction[1])

        def room(pos, step):                        # squares left on this axis before the ray leaves the board
            return board_size - 1 - pos if step > 0 else pos if step < 0 else board_size
        for k in range(1, min(room(row, dr), room(col, dc)) + 1):
            yield row + k * dr, col + k * dc

    @staticmethod
    def get_board_free_squares(board):
        return np.argwhere(~np.asarray(board, dtype=bool).any(axis=2))

    @staticmethod
    def is_board_square_free(board, row, col):
        return np.asarray(board[row, col]).max() == 0

    @staticmethod
    def _legal_mask(board, player):
        c0, c1 = _lib.pack_board(board)
        own, opp = (c0, c1) if player is OthelloPlayer.BLACK else (c1, c0)
        return int(rules_legal_moves([own], [opp], np.asarray(board).shape[0])[0])

    @staticmethod
    def get_player_valid_actions(board, player):
        return (np.array(a) for a in _mask_to_actions(OthelloGame._legal_mask(board, player)))

    @staticmethod
    def is_valid_player_action(board, player, row, col):
        return bool((OthelloGame._legal_mask(board, player) >> (int(row) * 8 + int(col))) & 1)

    @staticmethod
    def get_action_flip_squares(board, player, row, col):
        if not OthelloGame.is_board_square_free(board, row, col):
            return
        n = np.asarray(board).shape[0]
        c0, c1 = _lib.pack_board(board)
        own, opp = (c0, c1) if player is OthelloPlayer.BLACK else (c1, c0)
        o2, _ = rules_apply_moves([own], [opp], [int(row) * 8 + int(col)], n)
        flipped = int(o2[0]) & ~own & ~(1 << (int(row) * 8 + int(col)))
        yield from _mask_to_actions(flipped)

    @staticmethod
    def flip_board_squares(board, player, row, col):
        n = board.shape[0]
        c0, c1 = _lib.pack_board(board)
        own, opp = (c0, c1) if player is OthelloPlayer.BLACK else (c1, c0)
        o2, p2 = rules_apply_moves([own], [opp], [int(row) * 8 + int(col)], n)
        b, w = (int(o2[0]), int(p2[0])) if player is OthelloPlayer.BLACK else (int(p2[0]), int(o2[0]))
        board[...] = _lib.unpack_board(b, w, n)

    @staticmethod
    def has_board_finished(board):
        c0, c1 = _lib.pack_board(board)
        return bool(rules_status([c0], [c1], np.asarray(board).shape[0])[0][0])

    @staticmethod
    def get_board_winning_player(board):
        pts = OthelloGame.get_board_players_points(board)          # a draw goes to BLACK (first maximum, Othello:258-260)
        black, white = pts[OthelloPlayer.BLACK], pts[OthelloPlayer.WHITE]
        return (OthelloPlayer.BLACK, black) if black >= white else (OthelloPlayer.WHITE, white)

    @staticmethod
    def get_board_players_points(board):
        c0, c1 = _lib.pack_board(board)
        _, p0, p1, _ = rules_status([c0], [c1], np.asarray(board).shape[0])
        return {OthelloPlayer.BLACK: int(p0[0]), OthelloPlayer.WHITE: int(p1[0])}

    @staticmethod
    def has_player_actions_on_board(board, player):
        return OthelloGame._legal_mask(board, player) != 0

    @staticmethod
    def convert_to_one_channel_board(board):
        b = np.asarray(board)
        return b[:, :, 0].astype(np.int64) - b[:, :, 1].astype(np.int64)    # +1 BLACK, -1 WHITE, 0 empty (Othello:266-270)

    @staticmethod
    def invert_board(board):
        return np.asarray(board)[:, :, ::-1]          # a view with the two channels swapped, like np.flip(board, axis=2)
