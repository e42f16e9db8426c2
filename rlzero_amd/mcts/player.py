"""Player base classes (rlzero/mcts/player.py:5-57)."""


class Player(object):

    def __init__(self, player_id=0, player_name='') -> None:
        self.player_id = player_id
        self.player_name = player_name
        self.can_click = False

    def set_player_id(self, player_id):
        self.player_id = player_id

    def get_player_id(self):
        return self.player_id

    def get_player_name(self):
        return self.player_name

    def reset_player(self):
        raise NotImplementedError

    def get_action(self, game_env, **kwargs):
        raise NotImplementedError

    def __str__(self):
        return 'player'


class HumanPlayer(Player):
    """A player at the console (rlzero/mcts/player.py:33-57): asks for "row,col" until the answer names a legal move.  Host-only
    (no search, no GPU); ``ask`` replaces ``input`` in tests."""

    def __init__(self, player_id=0, player_name='', ask=None) -> None:
        super().__init__(player_id, player_name)
        self.can_click = True   # (the reference's GUI lets this player click the board)
        self._ask = ask if ask is not None else input

    def reset_player(self):
        pass

    def get_action(self, game_env, **kwargs):
        while True:
            move = -1
            try:
                text = self._ask('Your move: ')
                move = game_env.location_to_move([int(n, 10) for n in text.split(',')])
            except (ValueError, AttributeError, IndexError, TypeError) as exc:
                print(exc)
            if move != -1 and move in game_env.leagel_actions():
                return move
            print('invalid move')

    def __str__(self):
        return 'HumanPlayer, id: {}, name {}.'.format(self.get_player_id(), self.get_player_name())
