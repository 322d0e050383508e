"""OthelloGame -- drop-in for Othello/__init__.py:1-275 whose rule evaluation runs on the HIP rule kernels.

Same names, arguments, return types and error behaviour as the reference class; the board is kept as the
reference's (n,n,2) bool array (ch0 BLACK, ch1 WHITE) and mutated IN PLACE by play() (R10/R11), every
legal-move / flip / terminal query is one call into libothellozero_amd (batch of one).
Bulk work should use the batched entry points (`rules_*` below, `training.selfplay_batch`).
"""
from enum import Enum, auto

import numpy as np

from . import _lib


class BoardView(Enum):
    ONE_CHANNEL = auto()
    TWO_CHANNELS = auto()


class OthelloPlayer(Enum):
    BLACK = 1
    WHITE = -1

    @property
    def opponent(self):
        return OthelloPlayer.WHITE if self is OthelloPlayer.BLACK else OthelloPlayer.BLACK


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
        assert board_size % 2 == 0, 'Board size must be even'
        assert initial_board is None or initial_board.shape == (board_size, board_size, 2), \
            f'Expecting initial board shape ({board_size}, {board_size}, 2)'
        assert board_size in (4, 6, 8), 'the HIP rule kernels cover board sizes 4, 6 and 8'
        self._board = initial_board if initial_board is not None else self.initial_board(board_size)
        self._board_size = board_size
        self._round = 1
        self.current_player = current_player
        self._one_channel_board_last_update = None
        self._one_channel_board = None
        self._has_finished = OthelloGame.has_board_finished(self._board) if initial_board is not None else False

    @property
    def board_size(self):
        return self._board_size

    @property
    def round(self):
        return self._round

    def board(self, view=BoardView.ONE_CHANNEL):
        if view == BoardView.TWO_CHANNELS:
            return self._board                      # the live array, no copy (R11)
        elif view == BoardView.ONE_CHANNEL:
            if self._one_channel_board_last_update != self.round:
                self._one_channel_board = OthelloGame.convert_to_one_channel_board(self._board)
                self._one_channel_board_last_update = self.round
            return self._one_channel_board
        raise TypeError('Expecting BoardView type')

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
        n = self._board_size
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
        assert board_size % 2 == 0, 'Board size must be even'
        initial = np.array([[[0, 1], [1, 0]], [[1, 0], [0, 1]]], dtype=bool)
        pad = (board_size - 2) // 2
        return np.pad(initial, ((pad, pad), (pad, pad), (0, 0)), constant_values=0)

    @staticmethod
    def get_board_free_squares(board):
        return np.argwhere(np.amax(board, axis=2) == 0)

    @staticmethod
    def is_board_square_free(board, row, col):
        return np.amax(board[row, col]) == 0

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
        return max(OthelloGame.get_board_players_points(board).items(), key=lambda item: item[1])

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
        one_channel = board[:, :, 0] * OthelloPlayer.BLACK.value
        one_channel = one_channel + board[:, :, 1] * OthelloPlayer.WHITE.value
        return one_channel

    @staticmethod
    def invert_board(board):
        return np.flip(board, axis=2)
