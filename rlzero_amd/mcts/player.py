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
    """Reads "row,col" from stdin (interactive; not part of the accelerated path)."""

    def __init__(self, player_id=0, player_name=''):
        super().__init__(player_id, player_name)
        self.can_click = True

    def get_action(self, game_env, **kwargs):
        try:
            text = input('Your move: ')
            move = game_env.location_to_move([int(n, 10) for n in text.split(',')])
        except Exception as exc:  # noqa: BLE001
            print(exc)
            move = -1
        if move == -1 or move not in game_env.leagel_actions():
            print('invalid move')
            move = self.get_action(game_env)
        return move

    def __str__(self):
        return 'HumanPlayer, id: {}, name {}.'.format(self.get_player_id(), self.get_player_name())
